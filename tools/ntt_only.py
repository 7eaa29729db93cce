"""The 8 x 2^24 forward NTT step of bench.py on its own (BASELINE config 5 on one GPU), checked against the oracle's known
answers (tests/golden/ntt24_oracle.json) and timed with HIP events.  Used under rocprofv3 so that the kernel statistics and PMC
counters under profiles/ are NTT-only (tools/prof_ntt.sh).

    python tools/ntt_only.py [--cols 8] [--logn 24] [--steps 20] [--warmup 3] [--no-check]"""
import argparse
import ctypes
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from stark_brainfuck_amd import _lib  # noqa: E402
from stark_brainfuck_amd.device import DeviceBuffer, synchronize  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cols", type=int, default=8)
ap.add_argument("--logn", type=int, default=24)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--no-check", action="store_true")
ap.add_argument("--group", type=int, default=0, help="columns per bfs_gl_ntt call (0: all in one call)")
args = ap.parse_args()
lib = _lib.load()
n, cols = 1 << args.logn, args.cols
SEED = 0x5EED


def felt_array(seed, start, count):
    """splitmix64(seed + i) mod p"""
    with np.errstate(over="ignore"):
        x = (np.arange(start, start + count, dtype=np.uint64) + np.uint64(seed & 0xFFFFFFFFFFFFFFFF)) + np.uint64(0x9E3779B97F4A7C15)
        z = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    p = np.uint64(0xFFFFFFFF00000001)
    return np.where(z >= p, z - p, z)


src = DeviceBuffer(n * cols)
for c in range(cols):
    col = np.ascontiguousarray(felt_array(SEED + (c << 32), 0, n))
    _lib.check(lib.bfs_memcpy_h2d(src.ptr + 8 * c * n, col.ctypes.data, col.nbytes, 0))
    synchronize(0)
dst = DeviceBuffer(n * cols)
w = lib.bfs_gl_primitive_root(args.logn)


group = args.group or cols


def step():
    for c0 in range(0, cols, group):
        g = min(group, cols - c0)
        _lib.check(lib.bfs_gl_ntt(src.ptr + 8 * c0 * n, n, n, dst.ptr + 8 * c0 * n, n, args.logn, g, w, 1, 1, 0))


step()
synchronize(0)
checked = None
if not args.no_check and args.logn == 24:
    g = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ntt24_oracle.json")))
    for c in range(min(cols, 8)):
        got = hashlib.sha256(dst.to_numpy(n, offset=c * n).tobytes()).hexdigest()
        assert got == g["columns"][c]["output_sha256"], "column %d differs from the oracle's known answer" % c
    checked = min(cols, 8)
import time  # noqa: E402
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.4:          # clock spin-up: the first steps after idle run ~10 % slower
    step()
    synchronize(0)
for _ in range(args.warmup):
    step()
e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
lib.bfs_event_create(ctypes.byref(e0))
lib.bfs_event_create(ctypes.byref(e1))
lib.bfs_event_record(e0, 0)
for _ in range(args.steps):
    step()
lib.bfs_event_record(e1, 0)
ms = ctypes.c_float()
lib.bfs_event_elapsed_ms(e0, e1, ctypes.byref(ms))
per = ms.value / args.steps
print(json.dumps({"workload": "%d x 2^%d forward NTT" % (cols, args.logn), "columns_per_call": group,
                  "ms_per_step": round(per, 4), "elements_per_s": round(n * cols / per * 1e3), "algorithmic_GBps": round(16 * n * cols / per / 1e6, 1),
                  "roofline_frac_of_8TBps": round(16 * n * cols / per / 1e6 / 8000, 4), "columns_checked_vs_oracle": checked}))
