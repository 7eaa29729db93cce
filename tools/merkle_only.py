"""Merkle build over 2^log_n extension leaves, a few repetitions (profiling helper: rocprofv3 --kernel-trace --stats)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stark_brainfuck_amd import _lib
from stark_brainfuck_amd.device import DeviceBuffer
lib = _lib.load()
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 22
n = 1 << logn
src = DeviceBuffer.from_numpy(np.random.default_rng(1).integers(0, 2**63, 3 * n, dtype=np.uint64))
nodes = DeviceBuffer(2 * n * 8)
for _ in range(5):
    _lib.check(lib.bfs_merkle_build_xfe(src.ptr, n, n, nodes.ptr, 0))
_lib.check(lib.bfs_stream_synchronize(0))
