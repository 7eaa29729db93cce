"""Randomised parity soak on the MI355X: many sizes, roots, coset shifts, paddings and batch shapes against the CPU oracle
(oracle/, test infrastructure) plus device-vs-host primitive checks.  Not part of the test suite (minutes); prints a summary."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import ref_oracle as o
from stark_brainfuck_amd import _lib
from stark_brainfuck_amd.device import DeviceBuffer, synchronize
lib = _lib.load()
P = (1 << 64) - (1 << 32) + 1
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 12345)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
t_end = time.time() + budget
stats = {"ntt": 0, "selftest": 0, "merkle": 0, "fold": 0}
bad = ctypes.c_uint64(0)
_lib.check(lib.bfs_selftest_field(20, ctypes.byref(bad))); assert bad.value == 0; stats["selftest"] += 1
while time.time() < t_end:
    logn = int(rng.integers(0, 21))
    n = 1 << logn
    batch = int(rng.integers(1, 5)) if logn >= 12 else int(rng.integers(1, 40))
    # a random primitive n-th root: odd power of the canonical one
    w0 = o.primitive_nth_root(n) if n > 1 else 1
    root = o.power(w0, int(rng.integers(0, max(n // 2, 1))) * 2 + 1) if n > 1 else 1
    shift = int(rng.integers(1, P, dtype=np.uint64)) if rng.integers(0, 2) else 1
    scale = int(rng.integers(1, P, dtype=np.uint64)) if rng.integers(0, 2) else 1
    n_in = n if rng.integers(0, 2) else int(rng.integers(1, n + 1))
    in_stride = n_in + int(rng.integers(0, 3))
    out_stride = n + int(rng.integers(0, 3))
    v = rng.integers(0, P, in_stride * batch, dtype=np.uint64)
    if rng.integers(0, 4) == 0:
        v[rng.integers(0, 2, v.size) == 0] = 0
    if rng.integers(0, 2) == 0:
        # values next to 0, p and 2^32: what trace columns look like, and the only operands for which an unreduced sum of the butterfly
        # blocks really exceeds p (random ones do once in 2^32: round 4's first lazy-sum rule passed 5 922 random configurations here)
        small = rng.integers(0, 6, v.size, dtype=np.uint64)
        kind = rng.integers(0, 8, v.size)
        edge = np.where(kind < 3, small, np.uint64(P - 1) - small)
        edge = np.where(kind == 7, (np.uint64(1) << np.uint64(32)) - small, edge)
        v = np.where(kind == 6, v, edge).astype(np.uint64)
        stats["edge"] = stats.get("edge", 0) + 1
    din, dout = DeviceBuffer.from_numpy(v), DeviceBuffer(out_stride * batch)
    _lib.check(lib.bfs_gl_ntt(din.ptr, n_in, in_stride, dout.ptr, out_stride, logn, batch, root, shift, scale, 0))
    synchronize(0)
    got = dout.to_numpy()
    for b in range(batch):
        coeffs = v[b * in_stride:b * in_stride + n_in]
        want = o.fast_coset_evaluate(coeffs, shift, root, n) if n > 1 else coeffs.copy()
        if scale != 1:
            want = np.array([o.mul(int(x), scale) for x in want], dtype=np.uint64) if n <= 4096 else o.hadamard(want, np.full(n, scale, dtype=np.uint64))
        assert (got[b * out_stride:b * out_stride + n] == want).all(), ("ntt", logn, batch, root, shift, scale, n_in)
    stats["ntt"] += 1
    if rng.integers(0, 4) == 0:      # Merkle over a ragged extension codeword with special values
        m = int(rng.integers(1, 3000))
        soa = rng.integers(0, P, (3, m), dtype=np.uint64)
        for plane in (1, 2):
            soa[plane, rng.integers(0, 3, m) == 0] = 0
        buf = DeviceBuffer.from_numpy(soa.reshape(-1))
        npo2 = 1 << max((m - 1).bit_length(), 0)
        nodes = DeviceBuffer(2 * npo2 * 8)
        _lib.check(lib.bfs_merkle_build_xfe(buf.ptr, m, m, nodes.ptr, 0)); synchronize(0)
        tree, _ = o.xfe_merkle(soa)
        assert nodes.to_numpy(8, offset=8).tobytes() == tree.root(), ("merkle", m)
        stats["merkle"] += 1
print("soak ok:", stats, "seed", sys.argv[1] if len(sys.argv) > 1 else 12345)
