"""Start latency of a small operation after the GPU has idled for 0..200 ms (rules out power states as the cause of a late dispatch).
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stark_brainfuck_amd import _lib
from stark_brainfuck_amd.device import DeviceBuffer, synchronize
lib = _lib.load()
small = DeviceBuffer(1 << 10)
big = DeviceBuffer(1 << 27)       # 1 GiB
def burst(ms):
    t = time.perf_counter()
    while time.perf_counter() - t < ms * 1e-3:
        lib.bfs_memset(big.ptr, 0, big.nbytes, 0)
        synchronize(0)
for heavy in (0, 20):
    for idle in (0, 1, 2, 3, 4, 6, 8, 12, 20, 50, 200):
        lat = []
        for rep in range(5):
            if heavy: burst(heavy)
            synchronize(0)
            time.sleep(idle * 1e-3)
            t = time.perf_counter()
            lib.bfs_memset(small.ptr, 0, 1024, 0)
            synchronize(0)
            lat.append((time.perf_counter() - t) * 1e3)
        print("after %2d ms of load, idle %3d ms: small memset+sync latency %s ms" % (heavy, idle, " ".join("%.2f" % v for v in lat)), flush=True)
