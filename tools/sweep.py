"""Throughput sweep of the C-ABI entry points on one MI355X (development / DESIGN.md table)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stark_brainfuck_amd import _lib
from stark_brainfuck_amd.device import DeviceBuffer, synchronize
lib = _lib.load()


def timed(fn, reps=10):
    e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
    lib.bfs_event_create(ctypes.byref(e0)); lib.bfs_event_create(ctypes.byref(e1))
    for _ in range(2): fn()
    lib.bfs_event_record(e0, 0)
    for _ in range(reps): fn()
    lib.bfs_event_record(e1, 0)
    ms = ctypes.c_float(); lib.bfs_event_elapsed_ms(e0, e1, ctypes.byref(ms))
    return ms.value / reps


total = 1 << 27
src = DeviceBuffer.from_numpy(np.random.default_rng(1).integers(0, 2**63, total, dtype=np.uint64))
dst = DeviceBuffer(total)
print("NTT forward, 2^27 elements per call (batch = 2^27 / n):")
for logn in (10, 12, 13, 14, 16, 17, 18, 20, 22, 24, 26):
    n = 1 << logn
    batch = min(total // n, 65535)
    w = lib.bfs_gl_primitive_root(logn)
    ms = timed(lambda: _lib.check(lib.bfs_gl_ntt(src.ptr, n, n, dst.ptr, n, logn, batch, w, 1, 1, 0)))
    el = n * batch
    print("  2^%-2d x %-6d %8.3f ms  %6.1f G elem/s  %5.2f TB/s algorithmic" % (logn, batch, ms, el / ms / 1e6, 16 * el / ms / 1e9))
print("coset LDE (expansion 4): n_in = n/4, shift 7")
for logn in (20, 24):
    n = 1 << logn
    batch = total // n
    w = lib.bfs_gl_primitive_root(logn)
    ms = timed(lambda: _lib.check(lib.bfs_gl_ntt(src.ptr, n // 4, n // 4, dst.ptr, n, logn, batch, w, 7, 1, 0)))
    print("  2^%-2d x %-6d %8.3f ms  %6.1f G output elem/s" % (logn, batch, ms, n * batch / ms / 1e6))
print("Merkle over extension codewords (leaf = pickle of the element):")
for logn in (12, 16, 20, 22):
    n = 1 << logn
    nodes = DeviceBuffer(2 * n * 8)
    ms = timed(lambda: _lib.check(lib.bfs_merkle_build_xfe(src.ptr, n, n, nodes.ptr, 0)), reps=5)
    print("  N = 2^%-2d %8.3f ms  %6.1f M leaves/s  %6.1f GB/s algorithmic (152 N)" % (logn, ms, n / ms / 1e3, 152 * n / ms / 1e6))
print("fold:")
for logn in (20, 24):
    n = 1 << logn
    w = lib.bfs_gl_primitive_root(logn)
    ms = timed(lambda: _lib.check(lib.bfs_xfe_fold(src.ptr, n, dst.ptr, n // 2, logn, (ctypes.c_uint64 * 3)(5, 6, 7), 7, w, 0)))
    print("  N = 2^%-2d %8.3f ms  %6.1f GB/s algorithmic (36 N)" % (logn, ms, 36 * n / ms / 1e6))
