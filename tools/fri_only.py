"""Runs Fri.prove (config 3: d = 2^18, N = 2^20, expansion 4, t = 4) a few times through the C ABI; used under rocprofv3."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from stark_brainfuck_amd import _lib
lib = _lib.load()
r = bench.bench_fri(lib, _lib, 0, int(sys.argv[1]) if len(sys.argv) > 1 else 18)
print(r)
