"""Soak of the zero-padded transforms' expansion plans (ntt_plan.hpp: ntt_make_expand_plan) on the MI355X: a seeded sequence of
transforms whose coefficient counts sit around 2^M (M = 5..16, at most 1/16 of the domain; 0..16 coefficients past 2^M are the plan's
rank-one extras, 17 and more move it to the next M), ragged strides, random roots / coset offsets / scales and edge values.  The
process prints one SHA-256 per case; run it twice -- as it is and with BFS_NTT_EXPAND=0 (the dense plans, which tools/soak.py holds
against the CPU oracle) -- and compare the two listings: `tools/soak_expand.py SEED CASES > a; BFS_NTT_EXPAND=0 … > b; cmp a b`.
BFS_NTT_PLAN_LOG is not needed: the last line says how many cases took an expansion plan according to the planner's own rule."""
import ctypes, hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stark_brainfuck_amd import _lib
from stark_brainfuck_amd.device import DeviceBuffer, synchronize

lib = _lib.load()
P = (1 << 64) - (1 << 32) + 1
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
max_log = int(sys.argv[3]) if len(sys.argv) > 3 else 24
rng = np.random.default_rng(seed)


def root_of(log_n):
    # 7^((p-1)/2^log_n): the generator the library's own tables start from (field.py / csrc/field.hpp)
    return pow(7, (P - 1) >> log_n, P)


sparse = 0
for case in range(cases):
    log_n = int(rng.integers(9, max_log + 1))
    n = 1 << log_n
    M = int(rng.integers(5, min(16, log_n - 4) + 1))
    kind = int(rng.integers(0, 8))
    if kind < 3:
        n_in = (1 << M) + int(rng.integers(0, 17))
    elif kind == 3:
        n_in = (1 << M) + 17 + int(rng.integers(0, 4))
    elif kind == 4:
        n_in = (1 << M) - int(rng.integers(0, min(1 << M, 40)))
    elif kind == 5:
        n_in = int(rng.integers(1, (n >> 4) + 18))
    elif kind == 6:
        n_in = (1 << M) + 1
    else:
        n_in = int(rng.integers((1 << (M - 1)) + 1, (1 << M) + 1))
    n_in = max(1, min(n_in, n))
    sparse += n_in <= (n >> 4) + 16
    budget = 1 << 25
    batch = int(rng.integers(1, max(2, min(24, budget >> log_n) + 1)))
    w = root_of(log_n)
    root = pow(w, int(rng.integers(0, n // 2)) * 2 + 1, P) if rng.integers(0, 2) else w
    shift = int(rng.integers(1, P, dtype=np.uint64)) if rng.integers(0, 3) else 1
    scale = int(rng.integers(1, P, dtype=np.uint64)) if rng.integers(0, 3) == 0 else 1
    in_stride = n_in + int(rng.integers(0, 3))
    out_stride = n + int(rng.integers(0, 3))
    v = rng.integers(0, P, in_stride * batch, dtype=np.uint64)
    if rng.integers(0, 3) == 0:
        small = rng.integers(0, 6, v.size, dtype=np.uint64)
        k = rng.integers(0, 8, v.size)
        edge = np.where(k < 3, small, np.uint64(P - 1) - small)
        edge = np.where(k == 7, (np.uint64(1) << np.uint64(32)) - small, edge)
        v = np.where(k == 6, v, edge).astype(np.uint64)
    din, dout = DeviceBuffer.from_numpy(v), DeviceBuffer(out_stride * batch)
    _lib.check(lib.bfs_gl_ntt(din.ptr, n_in, in_stride, dout.ptr, out_stride, log_n, batch, root, shift, scale, 0))
    synchronize(0)
    got = dout.to_numpy().reshape(batch, out_stride)[:, :n]
    print(case, log_n, n_in, batch, int(shift != 1), int(scale != 1), hashlib.sha256(np.ascontiguousarray(got).tobytes()).hexdigest())
    del din, dout
print("cases", cases, "seed", seed, "with at most 1/16 of the domain filled:", sparse)
