"""Timing of the transforms the prover and FRI actually run: coset evaluations of few coefficients on a large domain (LDE) next to the
full 8 x 2^24 step.   python tools/lde_ab.py"""
import ctypes, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from stark_brainfuck_amd import _lib
from stark_brainfuck_amd.device import DeviceBuffer
lib = _lib.load()
def run(logn, n_in, cols, shift=7, steps=30):
    n = 1 << logn
    src = DeviceBuffer.from_numpy(np.random.default_rng(1).integers(0, 2**63, n_in * cols, dtype=np.uint64))
    dst = DeviceBuffer(n * cols)
    w = lib.bfs_gl_primitive_root(logn)
    f = lambda: _lib.check(lib.bfs_gl_ntt(src.ptr, n_in, n_in, dst.ptr, n, logn, cols, w, shift, 1, 0))
    for _ in range(5): f()
    e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
    lib.bfs_event_create(ctypes.byref(e0)); lib.bfs_event_create(ctypes.byref(e1))
    lib.bfs_event_record(e0, 0)
    for _ in range(steps): f()
    lib.bfs_event_record(e1, 0); _lib.check(lib.bfs_stream_synchronize(0))
    ms = ctypes.c_float(); lib.bfs_event_elapsed_ms(e0, e1, ctypes.byref(ms))
    return ms.value / steps
print("zero-padded and full transforms, HIP events over 30 calls each (development tool; A/B via BFS_LIB_PATH or the NTT environment switches)")
print("  LDE 16 cols 2^16+1 -> 2^22: %.3f ms" % run(22, (1 << 16) + 1, 16))
print("  LDE 27 cols 2^16+1 -> 2^22: %.3f ms" % run(22, (1 << 16) + 1, 27))
print("  xevaluate 3 x 2^22 -> 2^24: %.3f ms" % run(24, 1 << 22, 3))
print("  coset 8 x 2^22 -> 2^24:     %.3f ms" % run(24, 1 << 22, 8))
print("  full 8 x 2^24:              %.3f ms" % run(24, 1 << 24, 8, shift=1))
