"""Which buffers are slow?  (development tool)  The out-of-place passes of the 8 x 2^24 NTT step run at two speeds from one process to the
next on the same box (pass 0 447 or 472-495 us, pass 2 412 or 450-478 us; the in-place pass is always 439-447).  This times the step for
every ordered pair (input buffer, output buffer) out of K separately allocated 1 GiB buffers: if a BUFFER is slow, every pair that reads it
(pass 0) or writes it (pass 2) is slow.   python tools/buffer_pairs.py [K] [ws-first] [two-ws] [one-slab]
(ws-first: the library's intermediate buffer is allocated before the K buffers; two-ws: the table once more on a second stream, i.e. with
a second intermediate buffer.)  Findings: profiles/r03/buffer_placement.txt."""
import ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stark_brainfuck_amd import _lib
from stark_brainfuck_amd.device import DeviceBuffer, synchronize
lib = _lib.load()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
WS_FIRST = len(sys.argv) > 2 and sys.argv[2] == "ws-first"
n, cols, logn = 1 << 24, 8, 24
if WS_FIRST:                                   # a first transform allocates the library's intermediate buffer before the K buffers exist
    a, b = DeviceBuffer(n * cols), DeviceBuffer(n * cols)
    _lib.check(lib.bfs_gl_ntt(a.ptr, n, n, b.ptr, n, logn, cols, lib.bfs_gl_primitive_root(logn), 1, 1, 0))
    synchronize(0)
    print("first pair:", hex(a.ptr), hex(b.ptr), flush=True)
if "one-slab" in sys.argv:                     # the K buffers are consecutive 1 GiB ranges of ONE allocation
    slab = DeviceBuffer(n * cols * K)

    class View:
        def __init__(self, ptr):
            self.ptr = ptr
    bufs = [View(slab.ptr + i * n * cols * 8) for i in range(K)]
else:
    bufs = [DeviceBuffer(n * cols) for _ in range(K)]
print("addresses:", [hex(b.ptr) for b in bufs], flush=True)
v = (np.arange(n * cols, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) % np.uint64(0xFFFFFFFF00000001)
for b in bufs:
    _lib.check(lib.bfs_memcpy_h2d(b.ptr, v.ctypes.data, v.nbytes, 0))
synchronize(0)
w = lib.bfs_gl_primitive_root(logn)
e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
lib.bfs_event_create(ctypes.byref(e0)); lib.bfs_event_create(ctypes.byref(e1))


STREAM = None                                   # the library keeps one intermediate buffer per stream: a second stream = a second one


def run(i, j, steps):
    for _ in range(steps):
        _lib.check(lib.bfs_gl_ntt(bufs[i].ptr, n, n, bufs[j].ptr, n, logn, cols, w, 1, 1, STREAM))


t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.5:
    run(0, 1, 5); synchronize(0)
streams = [None]
if "two-ws" in sys.argv:
    h = ctypes.c_void_p()
    _lib.check(lib.bfs_stream_create(ctypes.byref(h)))
    streams.append(h)
for STREAM in streams:
  res = {}
  for rep in range(2):
      for i in range(K):
          for j in range(K):
              if i == j:
                  continue
              run(i, j, 2)
              lib.bfs_event_record(e0, STREAM)
              run(i, j, 12)
              lib.bfs_event_record(e1, STREAM)
              synchronize(STREAM)
              ms = ctypes.c_float(); lib.bfs_event_elapsed_ms(e0, e1, ctypes.byref(ms))
              res.setdefault((i, j), []).append(ms.value / 12)
  synchronize(STREAM)
  print("ms per step, rows = input buffer, columns = output buffer (two repetitions)")
  for i in range(K):
      print("  in %d: " % i + "  ".join("   --   " if i == j else "%.3f/%.3f" % tuple(res[(i, j)]) for j in range(K)))
