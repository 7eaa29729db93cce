"""Experiment: the 8 x 2^24 NTT step with its columns split over several HIP streams (fork / join per step with events), so that the
launch tails of one group are filled by the other group's kernels and memory-bound passes run next to VALU-bound ones.

    python tools/ntt_streams.py --splits 8 4,4 3,5 2,6 2,2,2,2 [--free]
    python tools/ntt_streams.py --percol 1 2 3 4      one call per COLUMN (batch 1: its 128 MiB stay in the 256 MiB Infinity Cache between
                                                      the passes), column c on stream c mod S: with S >= 2 the first pass of one column
                                                      overlaps the in-place passes of another and fills their launch tails (round-3 verdict #2)

--free: no fork / join between steps (every stream runs its own steps back to back): the upper bound of what overlap can give."""
import argparse
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from stark_brainfuck_amd import _lib  # noqa: E402
from stark_brainfuck_amd.device import DeviceBuffer, synchronize  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--logn", type=int, default=24)
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--splits", nargs="+", default=["8", "4,4", "3,5", "2,6", "2,2,2,2"])
ap.add_argument("--free", action="store_true")
ap.add_argument("--percol", nargs="*", type=int, default=None)
args = ap.parse_args()
lib = _lib.load()
hip = ctypes.CDLL("libamdhip64.so")
hip.hipStreamWaitEvent.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
hip.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
hip.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
n = 1 << args.logn
cols = 8
rng = np.random.default_rng(1)
src = DeviceBuffer.from_numpy(rng.integers(0, 0xFFFFFFFF00000001, size=n * cols, dtype=np.uint64))
dst = DeviceBuffer(n * cols)
w = lib.bfs_gl_primitive_root(args.logn)


def new_event():
    e = ctypes.c_void_p()
    assert hip.hipEventCreateWithFlags(ctypes.byref(e), 2) == 0      # hipEventDisableTiming
    return e


for S in (args.percol or []):
    streams = [ctypes.c_void_p(0)]
    for _ in range(S - 1):
        st = ctypes.c_void_p()
        _lib.check(lib.bfs_stream_create(ctypes.byref(st)))
        streams.append(st)
    fork = new_event()
    joins = [new_event() for _ in streams[1:]]

    def step():
        if S > 1:
            hip.hipEventRecord(fork, streams[0])
            for st in streams[1:]:
                hip.hipStreamWaitEvent(st, fork, 0)
        for c in range(cols):
            _lib.check(lib.bfs_gl_ntt(src.ptr + 8 * c * n, n, n, dst.ptr + 8 * c * n, n, args.logn, 1, w, 1, 1, streams[c % S]))
        if S > 1:
            for st, j in zip(streams[1:], joins):
                hip.hipEventRecord(j, st)
                hip.hipStreamWaitEvent(streams[0], j, 0)

    def sync():
        for st in streams:
            hip.hipStreamSynchronize(st)

    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.4:
        step()
        sync()
    best = None
    for rep in range(3):
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync()
        per = (time.perf_counter() - t0) / args.steps * 1e3
        best = per if best is None else min(best, per)
    print(json.dumps({"per_column_calls_on_streams": S, "streaming_env": os.environ.get("BFS_NTT_STREAMING"), "ms_per_step": round(best, 4),
                      "frac_of_8TBps": round(16 * n * cols / best / 1e6 / 8000, 4)}), flush=True)
    for st in streams[1:]:
        lib.bfs_stream_destroy(st)

for spec in (args.splits if args.percol is None else []):
    groups = [int(x) for x in spec.split(",")]
    assert sum(groups) == cols
    streams = [ctypes.c_void_p(0)]
    for _ in groups[1:]:
        s = ctypes.c_void_p()
        _lib.check(lib.bfs_stream_create(ctypes.byref(s)))
        streams.append(s)
    fork = new_event()
    joins = [new_event() for _ in groups[1:]]
    offs = [sum(groups[:i]) for i in range(len(groups))]

    def step():
        if len(groups) > 1 and not args.free:
            hip.hipEventRecord(fork, streams[0])
            for s in streams[1:]:
                hip.hipStreamWaitEvent(s, fork, 0)
        # the larger / later groups first on their side streams, the caller's stream last
        for g, o, s in list(zip(groups, offs, streams))[::-1]:
            _lib.check(lib.bfs_gl_ntt(src.ptr + 8 * o * n, n, n, dst.ptr + 8 * o * n, n, args.logn, g, w, 1, 1, s))
        if len(groups) > 1 and not args.free:
            for s, j in zip(streams[1:], joins):
                hip.hipEventRecord(j, s)
                hip.hipStreamWaitEvent(streams[0], j, 0)

    def sync():
        for s in streams:
            hip.hipStreamSynchronize(s)

    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.4:
        step()
        sync()
    best = None
    for rep in range(3):
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync()
        per = (time.perf_counter() - t0) / args.steps * 1e3
        best = per if best is None else min(best, per)
    print(json.dumps({"split": spec, "free_running": args.free, "ms_per_step": round(best, 4), "frac_of_8TBps": round(16 * n * cols / best / 1e6 / 8000, 4)}), flush=True)
    for s in streams[1:]:
        lib.bfs_stream_destroy(s)
