"""Host-to-device copy rates of DeviceBuffer.from_numpy for pageable and pinned sources, and the cost of a pooled allocate/free cycle.
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stark_brainfuck_amd import device
from stark_brainfuck_amd.device import DeviceBuffer
for mb in (1, 8, 32, 128):
    n = mb << 17
    parts = [np.arange(n // 4, dtype=np.uint64) for _ in range(4)]
    for rep in range(3):
        a = np.concatenate(parts)
        t = time.perf_counter(); b = DeviceBuffer.from_numpy(a); dt = time.perf_counter() - t
        back = b.to_numpy()
        assert (back == a).all()
        p = device.pinned_empty(n); p[:] = a
        t = time.perf_counter(); c = DeviceBuffer.from_numpy(p); dp = time.perf_counter() - t
        assert (c.to_numpy() == a).all()
        print("%4d MB  pageable %.2f ms (%.1f GB/s)   pinned %.2f ms (%.1f GB/s)" % (mb, dt * 1e3, mb / 1024 / dt, dp * 1e3, mb / 1024 / dp), flush=True)
        del b, c, p
t = time.perf_counter()
for i in range(200):
    b = DeviceBuffer(1 << 20)
print("alloc+free cycle %.1f us" % ((time.perf_counter() - t) / 200 * 1e6), device.pool_stats())
