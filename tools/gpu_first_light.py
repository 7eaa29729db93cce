"""First-light check on the GPU box: parity of bfs_gl_ntt vs the oracle over many sizes + a quick timing."""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stark_brainfuck_amd import _lib
from stark_brainfuck_amd.device import DeviceBuffer, synchronize
from oracle import ref_oracle as o

lib = _lib.load()


def gpu_ntt(v, logn, root, shift=1, scale=1, n_in=None, batch=1):
    n = 1 << logn
    n_in = n if n_in is None else n_in
    din = DeviceBuffer.from_numpy(v)
    dout = DeviceBuffer(n * batch)
    _lib.check(lib.bfs_gl_ntt(din.ptr, n_in, n_in, dout.ptr, n, logn, batch, root, shift, scale, 0))
    synchronize(0)
    return dout.to_numpy()


bad = 0
for logn in range(0, 23):
    n = 1 << logn
    v = o.felt_array(0x5EED, 0, n)
    w = o.primitive_nth_root(n)
    t0 = time.time()
    ok1 = (gpu_ntt(v, logn, w) == o.ntt(w, v)).all()
    ok2 = (gpu_ntt(v, logn, o.inv(w), 1, o.inv(n)) == o.intt(w, v)).all()
    d = max(1, n // 4)
    ok3 = (gpu_ntt(v[:d], logn, w, 7, 1, n_in=d) == o.fast_coset_evaluate(v[:d], 7, w, n)).all()
    print(logn, ok1, ok2, ok3, "%.2fs" % (time.time() - t0), flush=True)
    bad += (not ok1) + (not ok2) + (not ok3)
print("MISMATCHES", bad)

# timing: 2^24 x 8 columns, HIP events
for logn, batch in ((20, 8), (24, 1), (24, 8)):
    n = 1 << logn
    w = o.primitive_nth_root(n)
    din = DeviceBuffer.from_numpy(o.felt_array(1, 0, n * batch))
    dout = DeviceBuffer(n * batch)
    e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
    lib.bfs_event_create(ctypes.byref(e0)); lib.bfs_event_create(ctypes.byref(e1))
    for _ in range(3):
        _lib.check(lib.bfs_gl_ntt(din.ptr, n, n, dout.ptr, n, logn, batch, w, 1, 1, 0))
    synchronize(0)
    reps = 10
    lib.bfs_event_record(e0, 0)
    for _ in range(reps):
        _lib.check(lib.bfs_gl_ntt(din.ptr, n, n, dout.ptr, n, logn, batch, w, 1, 1, 0))
    lib.bfs_event_record(e1, 0)
    ms = ctypes.c_float()
    lib.bfs_event_elapsed_ms(e0, e1, ctypes.byref(ms))
    t = ms.value / reps * 1e-3
    print("logn %d batch %d: %.3f ms  %.2f Gelem/s  %.1f GB/s algorithmic" % (logn, batch, t * 1e3, n * batch / t / 1e9, 16 * n * batch / t / 1e9), flush=True)
