"""CPU checks of the product's DEVICE code compiled for the host (tests/emu/*.cpp): the NTT planner and every
index / twiddle computation of the tile kernels, the pickle-exact leaf encoder, BLAKE2b, SHAKE256 and the Merkle
bodies -- against the oracle and the reference goldens.  The emulation library is test infrastructure; the product
never executes these paths on the CPU."""
import ctypes
import hashlib
import os

import numpy as np
import pytest

from conftest import ROOT, load_golden

SEED = 0x5EED
P = (1 << 64) - (1 << 32) + 1
EMU_DIR = os.path.join(ROOT, "tests", "emu")
u64, vp = ctypes.c_uint64, ctypes.c_void_p


@pytest.fixture(scope="session")
def emu():
    import sys
    sys.path.insert(0, EMU_DIR)
    from build_emu import build_emulation
    so = build_emulation()          # rebuilt whenever the CONTENT of a source or header differs from what the .so was made of
    lib = ctypes.CDLL(so)
    lib.emu_gl_ntt.argtypes = [vp, u64, u64, vp, u64, ctypes.c_uint32, ctypes.c_uint32, u64, u64, u64]
    lib.emu_set_schedule.argtypes = [ctypes.c_int, ctypes.c_uint32, u64]
    lib.emu_set_force_ws.argtypes = [ctypes.c_int]
    lib.emu_set_expand.argtypes = [ctypes.c_int]
    lib.emu_set_expand.restype = ctypes.c_ulonglong
    lib.emu_canonical_violations.argtypes = [ctypes.c_int]
    lib.emu_canonical_violations.restype = ctypes.c_ulonglong
    lib.emu_plan.argtypes = [ctypes.c_uint32, u64, vp, vp, vp, vp]
    lib.emu_merkle_xfe.argtypes = [vp, u64, u64, vp]
    lib.emu_xfe_leaf_stream.argtypes = [vp, u64, u64, ctypes.c_int, vp]
    u32 = ctypes.c_uint32
    lib.emu_row_leaves.argtypes = [vp, vp, vp, u32, vp, u32, vp, u32, u32, u32, vp, vp]
    return lib


def emu_ntt(lib, v, logn, root, shift=1, scale=1, n_in=None, batch=1):
    n = 1 << logn
    v = np.ascontiguousarray(v, dtype=np.uint64)
    n_in = n if n_in is None else n_in
    out = np.zeros(n * batch, dtype=np.uint64)
    lib.emu_canonical_violations(1)
    rc = lib.emu_gl_ntt(v.ctypes.data, n_in, n_in, out.ctypes.data, n, logn, batch, root, shift, scale)
    assert rc == 0, rc
    # the emulation is built with -DBFS_CHECK_CANONICAL: every operand that must be a canonical residue was one (gl.hpp)
    assert lib.emu_canonical_violations(1) == 0
    return out


def edge_values(n, seed):
    """values next to 0 and next to p, a few random ones in between: an unreduced sum (gl_add_lazy) is only really >= p when it
    lands within 2^32 of p, which random operands do once in 2^32 and these do all the time -- the trace columns of a proof look like
    this (small counters, 0, p - 1), and a proof is where a non-canonical value slipped through in round 4 (tools/soak_stark.py)"""
    rng = np.random.default_rng(seed)
    small = rng.integers(0, 6, n, dtype=np.uint64)
    kind = rng.integers(0, 8, n)
    v = np.where(kind < 3, small, np.uint64(P) - np.uint64(1) - small)
    v = np.where(kind == 6, rng.integers(0, P, n, dtype=np.uint64), v)
    v = np.where(kind == 7, (np.uint64(1) << np.uint64(32)) - small, v)            # around EPS
    return np.ascontiguousarray(v, dtype=np.uint64)


@pytest.mark.parametrize("logn", list(range(0, 15)) + [16, 17, 18, 20])
def test_tile_kernels_match_oracle(emu, oracle, logn):
    n = 1 << logn
    v = oracle.felt_array(SEED, 0, n)
    w = oracle.primitive_nth_root(n)
    assert (emu_ntt(emu, v, logn, w) == oracle.ntt(w, v)).all()
    assert (emu_ntt(emu, v, logn, oracle.inv(w), 1, oracle.inv(n)) == oracle.intt(w, v)).all()
    d = max(1, n // 4)
    assert (emu_ntt(emu, v[:d], logn, w, 7, 1, n_in=d) == oracle.fast_coset_evaluate(v[:d], 7, w, n)).all()


@pytest.mark.parametrize("logn", list(range(1, 15)) + [16, 17, 18])
def test_values_next_to_zero_and_p_stay_canonical(emu, oracle, logn):
    """every plan shape on operands that make unreduced sums actually exceed p: the oracle's transform, bit for bit, and no
    non-canonical operand anywhere a canonical one is required (emu_ntt asserts the emulation's violation count)"""
    n = 1 << logn
    w = oracle.primitive_nth_root(n)
    for seed in range(3 if logn <= 14 else 1):
        v = edge_values(n, 1000 * logn + seed)
        assert (emu_ntt(emu, v, logn, w) == oracle.ntt(w, v)).all(), seed
        assert (emu_ntt(emu, v, logn, oracle.inv(w), 1, oracle.inv(n)) == oracle.intt(w, v)).all(), seed
        d = max(1, n // 4)
        assert (emu_ntt(emu, v[:d], logn, w, 7, 1, n_in=d) == oracle.fast_coset_evaluate(v[:d], 7, w, n)).all(), seed
        assert (emu_ntt(emu, v[:d], logn, w, 1, 1, n_in=d) == oracle.fast_coset_evaluate(v[:d], 1, w, n)).all(), seed


@pytest.mark.parametrize("logn", [17, 18, 19, 20])
def test_three_pass_plans_under_both_twiddle_schedules(emu, oracle, logn):
    """the balanced schedule (store-time row in pass 1, one factor per thread in pass 2, load-time row in pass 3) and the load-time
    schedule (row table in pass 2, chain in pass 3) give the oracle's transform: forward, inverse with n^-1, coset with zero padding"""
    n = 1 << logn
    v = oracle.felt_array(SEED + logn, 0, n)
    w = oracle.primitive_nth_root(n)
    want = oracle.ntt(w, v)
    want_inv = oracle.intt(w, v)
    d = n // 4
    want_coset = oracle.fast_coset_evaluate(v[:d], 7, w, n)
    used = []
    try:
        for mode in (0, 1):
            used.append(emu.emu_set_schedule(mode, logn, w))
            assert (emu_ntt(emu, v, logn, w) == want).all(), mode
            assert (emu_ntt(emu, v, logn, oracle.inv(w), 1, oracle.inv(n)) == want_inv).all(), mode
            assert (emu_ntt(emu, v[:d], logn, w, 7, 1, n_in=d) == want_coset).all(), mode
    finally:
        emu.emu_set_schedule(-1, logn, w)
    assert used[0] == 0
    assert used[1] == 1 or logn == 17          # 2^17 = 6 + 6 + 5 (or 5 + 6 + 6): a first-pass tile spans two values of the next digit


@pytest.mark.parametrize("logn", [4, 7, 10, 12, 13, 16, 18, 20])
def test_zero_padded_inputs_of_every_shape(emu, oracle, logn):
    """fast_coset_evaluate of few coefficients on a large domain (ntt.py:164-168; every trace column of the prover: h + 1 coefficients
    on 64 h points): coefficient counts around n / 64, n / 16, n / 4 and n / 2, with and without the coset shift -- the first pass'
    loads are predicated on n_in and everything above reads as zero."""
    n = 1 << logn
    w = oracle.primitive_nth_root(n)
    counts = sorted({1, 2, max(1, n // 64), max(1, n // 64) + 1, n // 16, n // 16 + 1, n // 8, n // 8 + 1, n // 4 - 1, n // 4, n // 4 + 1, n // 2, n // 2 + 3} & set(range(1, n + 1)))
    v = oracle.felt_array(SEED + 3 * logn, 0, n)
    for d in counts:
        for shift in (1, 7):
            want = oracle.fast_coset_evaluate(v[:d], shift, w, n)
            assert (emu_ntt(emu, v[:d], logn, w, shift, 1, n_in=d) == want).all(), (logn, d, shift)


@pytest.mark.parametrize("logn", [13, 14, 16, 17, 18, 20])
def test_expansion_plans_of_zero_padded_transforms(emu, oracle, logn):
    """ntt_make_expand_plan / PASS_EXPAND: when the coefficients fill at most 1/16 of the domain the plan starts at the second digit of the
    input index -- one real pass for up to 2^8 (+ extras) coefficients, two for up to 2^16 -- and the first real pass reads the coefficients
    themselves, carries the coset shift and adds the rank-one terms of the few coefficients past the power of two (a trace interpolant has
    height + 1).  Every count around the plan's thresholds, with and without shift and post-scale, must equal the oracle's
    fast_coset_evaluate (ntt.py:164-168) and the plain plan's output, and the plan must actually have been taken."""
    n = 1 << logn
    w = oracle.primitive_nth_root(n)
    v = oracle.felt_array(SEED + 5 * logn, 0, n)
    P = (1 << 64) - (1 << 32) + 1
    edge = np.array([0, P - 1, 1, P - 2] * (n // 4), dtype=np.uint64)
    taken = 0
    counts = set()
    for M in range(3, min(16, logn - 4) + 1):
        counts |= {(1 << M) - 1, 1 << M, (1 << M) + 1, (1 << M) + 2, (1 << M) + 16, (1 << M) + 17}
    counts = sorted(c for c in counts if 1 <= c <= n // 16 + 17)
    if logn >= 18:                      # (the host emulation walks every tile of every pass: keep the large sizes to the counts next to n / 64 .. n / 16)
        counts = [c for c in counts if c >= n // 128]
    if logn >= 20:
        counts = [c for c in counts if c in (n // 64 + 1, n // 32, n // 16 + 16, n // 16 + 17)]
    for d in counts:
        for shift, scale, src in ((7, 1, v), (1, 1, edge), (3, oracle.inv(n), v)):
            before = emu.emu_set_expand(1)
            got = emu_ntt(emu, src[:d], logn, w, shift, scale, n_in=d)
            after = emu.emu_set_expand(0)
            plain = emu_ntt(emu, src[:d], logn, w, shift, scale, n_in=d)
            emu.emu_set_expand(1)
            taken += after - before
            want = oracle.fast_coset_evaluate(src[:d], shift, w, n)
            if scale != 1:
                want = oracle.mul_scalar(want, scale) if hasattr(oracle, "mul_scalar") else plain
            assert (got == want).all() and (got == plain).all(), (logn, d, shift, scale)
    assert taken >= len(counts), (taken, len(counts))


@pytest.mark.parametrize("logn", [13, 16, 17, 20])
def test_multi_pass_buffer_flow(emu, oracle, logn):
    """round 4: pass 0 transposes from the input into the output and every later pass runs in place there; when input and output
    overlap (a caller transforming in place) passes 0 and 1 go through the intermediate buffer instead.  Same values either way, for
    a batch whose transforms are spaced wider than n on both sides."""
    n = 1 << logn
    w = oracle.primitive_nth_root(n)
    batch, in_stride, out_stride = 2, n + 24, n + 8
    src = np.zeros(in_stride * batch, dtype=np.uint64)
    cols = [oracle.felt_array(SEED + 77 + b, 0, n) for b in range(batch)]
    for b in range(batch):
        src[b * in_stride:b * in_stride + n] = cols[b]
    want = [oracle.ntt(w, c) for c in cols]

    def run(buf_in, buf_out, stride_in, stride_out):
        rc = emu.emu_gl_ntt(buf_in.ctypes.data, n, stride_in, buf_out.ctypes.data, stride_out, logn, batch, w, 1, 1)
        assert rc == 0

    out = np.zeros(out_stride * batch, dtype=np.uint64)
    run(src, out, in_stride, out_stride)                          # separate buffers: no intermediate buffer
    for b in range(batch):
        assert (out[b * out_stride:b * out_stride + n] == want[b]).all()
        assert not out[b * out_stride + n:(b + 1) * out_stride].any()          # nothing written between the transforms
    emu.emu_set_force_ws(1)
    try:
        out2 = np.zeros(out_stride * batch, dtype=np.uint64)
        run(src, out2, in_stride, out_stride)                     # the overlapping-buffers route, forced
        assert (out2 == out).all()
    finally:
        emu.emu_set_force_ws(0)
    inplace = src.copy()
    run(inplace, inplace, in_stride, in_stride)                   # in == out
    for b in range(batch):
        assert (inplace[b * in_stride:b * in_stride + n] == want[b]).all()


def test_tile_kernels_batch_and_other_roots(emu, oracle):
    logn, n = 13, 1 << 13
    w = oracle.power(oracle.primitive_nth_root(n), 5)          # another primitive root (odd power)
    v = oracle.felt_array(SEED + 9, 0, 3 * n)
    got = emu_ntt(emu, v, logn, w, batch=3).reshape(3, n)
    for b in range(3):
        assert (got[b] == oracle.ntt(w, v[b * n:(b + 1) * n])).all()


def test_golden_ntt_through_emulation(emu):
    c = load_golden("ntt.json")["cases"]["10"]
    import oracle.ref_oracle as o
    v = o.felt_array(SEED, 0, 1024)
    assert emu_ntt(emu, v, 10, c["root"]).tolist() == c["ntt"]


def test_planner(emu, oracle):
    for logn in range(4, 33):
        npass, uinv = ctypes.c_uint32(), ctypes.c_uint32()
        bits, logc = (ctypes.c_uint32 * 4)(), (ctypes.c_uint32 * 4)()
        root = oracle.primitive_nth_root(1 << logn)
        assert emu.emu_plan(logn, root, ctypes.byref(npass), bits, logc, ctypes.byref(uinv)) == 0
        assert sum(bits[:npass.value]) == logn
        if npass.value > 1:
            assert all(5 <= b <= 8 for b in bits[:npass.value])
            assert all(bits[i] + logc[i] == 12 for i in range(npass.value))
        # the radix-16 root of the reference is 2^(12u): u * uinv == 1 (mod 16)
        w16 = oracle.power(root, (1 << logn) // 16)
        u = [k for k in range(1, 16, 2) if pow(2, 12 * k, oracle.P) == w16][0]
        assert (u * uinv.value) % 16 == 1


def test_leaf_encoder_matches_reference_pickles(emu):
    g = load_golden("pickle.json")
    for r in g["xfe_leaves"]:
        l = r["limbs"] + [0] * (3 - len(r["limbs"]))
        buf = ctypes.create_string_buffer(424)
        n = emu.emu_xfe_leaf_pickle((u64 * 3)(*l), buf)
        assert buf.raw[:n].hex() == r["pickle"], r["limbs"]
    for r in g["bfe_leaves"]:
        buf = ctypes.create_string_buffer(136)
        n = emu.emu_bfe_leaf_pickle(u64(r["limbs"][0]), buf)
        assert buf.raw[:n].hex() == r["pickle"]
        d = ctypes.create_string_buffer(64)
        emu.emu_blake2b(buf.raw[:n], ctypes.c_size_t(n), d)
        assert d.raw.hex() == r["blake2b"]


def test_hashes_match_hashlib(emu):
    for ln in [0, 1, 63, 64, 127, 128, 129, 255, 256, 300, 409, 1000]:
        data = bytes((i * 7 + 3) & 255 for i in range(ln))
        d = ctypes.create_string_buffer(64)
        emu.emu_blake2b(data, ctypes.c_size_t(ln), d)
        assert d.raw == hashlib.blake2b(data).digest()
        for ol in (32, 136, 200):
            o = ctypes.create_string_buffer(ol)
            emu.emu_shake256(data, ctypes.c_size_t(ln), o, ctypes.c_size_t(ol))
            assert o.raw == hashlib.shake_256(data).digest(ol)


@pytest.mark.parametrize("klass", [1, 2, 3])
def test_streamed_leaf_equals_hashlib_over_the_reference_pickle(emu, klass):
    """merkle_leaf_xfe_stream<K> (first tail block compressed in the middle of the encoding, 20 words of staging per lane) against
    BLAKE2b of the whole pickle -- the encoder itself is pinned on the reference's pickles above.  Coefficients of every pickle integer
    length (2, 3, 5 and 3..11 bytes), so the lanes' positions at the split differ by the full 27 bytes; stale bytes in the staging area."""
    rng = np.random.default_rng(40 + klass)
    sizes = [1, 7, 8, 15, 16, 30, 31, 32, 33, 40, 47, 48, 55, 56, 57, 63, 64]

    def value(bits):
        v = int(rng.integers(1 << (bits - 1), (1 << bits) - 1, dtype=np.uint64)) if bits > 1 else 1
        return v % ((1 << 64) - (1 << 32) + 1) or 1
    n = 64 * 12
    limbs = np.zeros((3, n), dtype=np.uint64)
    for i in range(n):
        for l in range(klass):
            limbs[l, i] = value(sizes[int(rng.integers(0, len(sizes)))]) if (l == klass - 1 or rng.integers(0, 6)) else 0
        assert limbs[klass - 1, i] != 0
    digests = np.zeros(8 * n, dtype=np.uint64)
    assert emu.emu_xfe_leaf_stream(limbs.ctypes.data, n, n, klass, digests.ctypes.data) == 0
    for i in range(n):
        buf = ctypes.create_string_buffer(424)
        ln = emu.emu_xfe_leaf_pickle((u64 * 3)(*[int(limbs[l, i]) for l in range(3)]), buf)
        assert digests[8 * i:8 * i + 8].tobytes() == hashlib.blake2b(buf.raw[:ln]).digest(), (klass, i, [int(limbs[l, i]) for l in range(3)])


def test_merkle_bodies_match_reference_trees(emu, oracle):
    for t in load_golden("merkle.json")["xfe_trees"]:
        n = t["n"]
        soa = np.array([[oracle.felt(SEED + t["seed_offset"], 3 * i + k) for i in range(n)] for k in range(3)], dtype=np.uint64)
        npo2 = 1
        while npo2 < n:
            npo2 *= 2
        nodes = np.zeros(2 * npo2 * 8, dtype=np.uint64)
        emu.emu_merkle_xfe(soa.ctypes.data, n, n, nodes.ctypes.data)
        got = [nodes[i * 8:(i + 1) * 8].tobytes().hex() for i in range(2 * npo2)]
        for i in range(1, npo2 + n):
            assert got[i] == t["nodes"][i], (n, i)


# ------------------------------------------------------------------ zipped-row leaf kernel: the per-lane state machine
def _pickle_int(v):
    """CPython's save_long for a non-negative int below 2^64 (what pickle protocol 4 writes for the reference's element values)"""
    if v < 1 << 8:
        return b"K" + bytes([v])
    if v < 1 << 16:
        return b"M" + v.to_bytes(2, "little")
    if v < 1 << 31:
        return b"J" + v.to_bytes(4, "little")
    nn = v.bit_length() // 8 + 1
    return b"\x8a" + bytes([nn]) + v.to_bytes(nn, "little")


@pytest.mark.parametrize("seed", range(12))
def test_row_lanes_walk_a_template_block_synchronously(emu, seed):
    """csrc/rows_core.hpp on the host: 64 lanes walk a random flattened template (constant runs, integers of every opcode width,
    frame length, salt) in lockstep, with unaligned 8-byte stores into ROW_LANE_BYTES-byte lane buffers full of stale bytes and wave-wide
    compressions; every lane's digest must be BLAKE2b of the bytes a direct construction gives.  Value mixes: all nine-byte
    integers (lanes drift by a byte or two), random widths, and alternating all-small / all-large rows (lanes whole blocks apart,
    which sit compressions out); lengths that end on block boundaries occur along the way."""
    import pickle
    rng = np.random.default_rng(1000 + seed)
    lanes, PAD = 64, 4
    SEG_CONST, SEG_INT, SEG_FRAMELEN, SEG_SALT, SEG_INT_HI = 0, 1, 2, 3, 4
    pieces = [("c", bytes(rng.integers(0, 256, 3, dtype=np.uint8))), ("f", None)]
    nints = 0
    for _ in range(int(rng.integers(1, 70))):
        if rng.integers(0, 3) == 0:
            pieces.append(("c", bytes(rng.integers(0, 256, int(rng.integers(1, 41)), dtype=np.uint8))))
        else:
            pieces.append(("i", nints))
            nints += 1
    salted = bool(seed % 2)
    salt_pickle_pre, salt_pickle_post = b"\x80\x04\x95\x1c\x00\x00\x00\x00\x00\x00\x00C\x18", b"\x94."
    assert pickle.dumps(bytes(24), protocol=4) == salt_pickle_pre + bytes(24) + salt_pickle_post
    kinds, As, datas = [], [], []

    def add_const(b):
        for w in range(0, len(b), 8):
            chunk = b[w:w + 8]
            kinds.append(SEG_CONST); As.append(len(chunk)); datas.append(int.from_bytes(chunk, "little"))
    const_bytes = 0
    for kind, arg in pieces:
        if kind == "c":
            add_const(arg); const_bytes += len(arg)
        elif kind == "f":
            kinds.append(SEG_FRAMELEN); As.append(0); datas.append(0); const_bytes += 8
        else:
            kinds += [SEG_INT, SEG_INT_HI]; As += [arg, arg]; datas += [0, 0]
    if salted:
        add_const(salt_pickle_pre)
        for w in range(3):
            kinds.append(SEG_SALT); As.append(w); datas.append(0)
        add_const(salt_pickle_post)
    while len(kinds) % PAD:
        kinds.append(SEG_CONST); As.append(0); datas.append(0)
    P = (1 << 64) - (1 << 32) + 1
    widths = [0, 255, 256, 65535, 65536, (1 << 31) - 1, 1 << 31, (1 << 40) - 1, 1 << 47, (1 << 48) + 9, (1 << 56) - 1, 1 << 56, (1 << 63) - 1, 1 << 63, P - 1]
    mode = seed % 3
    values = np.zeros((lanes, max(nints, 1)), dtype=np.uint64)
    for l in range(lanes):
        for j in range(nints):
            if mode == 0:
                values[l, j] = int(rng.integers(1 << 63, P, dtype=np.uint64))
            elif mode == 1:
                values[l, j] = widths[int(rng.integers(0, len(widths)))]
            else:
                values[l, j] = 5 if l % 2 else P - 1 - j
    salts = rng.integers(0, 1 << 63, (lanes, 3), dtype=np.uint64)
    k32 = np.array(kinds, dtype=np.uint32); a32 = np.array(As, dtype=np.uint32); d64 = np.array(datas, dtype=np.uint64)
    out = np.zeros((lanes, 8), dtype=np.uint64)
    skew = ctypes.c_uint32()
    vals = np.ascontiguousarray(values[:, :nints]) if nints else np.zeros(1, dtype=np.uint64)
    rc = emu.emu_row_leaves(k32.ctypes.data, a32.ctypes.data, d64.ctypes.data, len(kinds), vals.ctypes.data, nints, salts.ctypes.data,
                            const_bytes, len(salt_pickle_pre) + 24 + len(salt_pickle_post) if salted else 0, lanes, out.ctypes.data, ctypes.byref(skew))
    assert rc == 0, rc
    for l in range(lanes):
        ints = [_pickle_int(int(values[l, j])) for j in range(nints)]
        tuple_len = const_bytes + sum(len(b) for b in ints)
        pre = b""
        for kind, arg in pieces:
            pre += arg if kind == "c" else ((tuple_len - 11).to_bytes(8, "little") if kind == "f" else ints[arg])
        assert len(pre) == tuple_len
        if salted:
            pre += salt_pickle_pre + salts[l].tobytes() + salt_pickle_post
        assert out[l].tobytes() == hashlib.blake2b(pre).digest(), "lane %d, %d bytes" % (l, len(pre))
    if mode == 2 and nints >= 8:
        assert skew.value > 56, "the alternating rows should put lanes more than a buffer's slack apart"


def test_generated_row_header_is_what_the_generator_writes_today():
    """csrc/rows_generated.hpp against tools/gen_rows.py (which asks the library for the templates: bfs_row_template_steps)"""
    import subprocess
    import sys
    from stark_brainfuck_amd import build
    build.build_library()                 # (no-op when current: the generator reads the templates through the library)
    header = os.path.join(ROOT, "stark_brainfuck_amd", "csrc", "rows_generated.hpp")
    before = open(header).read()
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rows.py")], check=True, capture_output=True)
    assert open(header).read() == before, "rows_generated.hpp is stale: run tools/gen_rows.py"


@pytest.mark.parametrize("layout,variant,n_ext,n_base", [(0, 0, 1, 16)] + [(1, v, 9, 0) for v in range(9)])
def test_generated_row_walks_against_the_oracle_pickles(emu, oracle, layout, variant, n_ext, n_base):
    """csrc/rows_generated.hpp (tools/gen_rows.py: the zipped-row template walk unrolled for the prover's two column layouts and the
    row patterns they come in), compiled for the host and driven like row_leaves_generated_kernel drives it (tests/emu/emu_rowgen.cpp):
    64 lanes in lockstep, every digest against BLAKE2b of the oracle's pickle of the same row (look-alike classes, CPython's pickle:
    brainfuck_stark.py:178-179 / 197-198 with salted_merkle.py:32-35).  Integers of every opcode width, neighbouring lanes blocks
    apart.  Also: the library matches its own template of the pattern by the hash the header carries."""
    u32 = ctypes.c_uint32
    emu.emu_rowgen_leaves.argtypes = [u32, u32, vp, vp, u32, vp, vp, vp]
    info = (u32 * 5)()
    assert emu.emu_rowgen_leaves(layout, variant, None, None, 0, None, info, None) == 0
    nints, code = info[1], info[4]
    stored = [(code >> (2 * e)) & 3 for e in range(n_ext)]        # coefficients per extension element
    assert nints == sum(stored) + n_base
    lanes = 64
    widths = [1, 255, 256, 65535, 65536, (1 << 31) - 1, 1 << 31, (1 << 40) - 1, 1 << 47, (1 << 48) + 9, (1 << 56) - 1, 1 << 56, (1 << 63) - 1, 1 << 63, P - 1]
    for mode in range(3):
        rng = np.random.default_rng(SEED + 100 * layout + 10 * variant + mode)
        values = np.zeros((lanes, nints), dtype=np.uint64)
        for l in range(lanes):
            for j in range(nints):
                if mode == 0:
                    values[l, j] = int(rng.integers(1, P, dtype=np.uint64))
                elif mode == 1:
                    values[l, j] = widths[int(rng.integers(0, len(widths)))]
                else:
                    values[l, j] = 5 if l % 2 else P - 1 - j
        salts = rng.integers(0, 1 << 63, (lanes, 3), dtype=np.uint64)
        out = np.zeros((lanes, 8), dtype=np.uint64)
        sites = u32()
        rc = emu.emu_rowgen_leaves(layout, variant, values.ctypes.data, salts.ctypes.data, lanes, out.ctypes.data, None, ctypes.byref(sites))
        assert rc == 0, rc
        longest = 0
        for l in range(lanes):
            v = [int(x) for x in values[l]]
            row, at = [], 0
            for k in stored:
                row.append(oracle.make_xfe(v[at:at + k] + [0] * (3 - k)))
                at += k
            row += [oracle.make_bfe(x) for x in v[at:]]
            pre = oracle.salted_leaf_bytes(tuple(row), salts[l].tobytes())
            longest = max(longest, len(pre))
            assert out[l].tobytes() == hashlib.blake2b(pre).digest(), "mode %d lane %d, %d bytes" % (mode, l, len(pre))
        blocks_walked = (longest + 127) // 128 - 1              # (block 0 is a table look-up, not a compression of the walk)
        assert sites.value >= blocks_walked
        if mode == 2:
            assert sites.value > blocks_walked, "lanes blocks apart: some compressions must run without the short rows"
    from stark_brainfuck_amd import _lib, build
    build.build_library()
    lib = _lib.load()
    cols = (_lib.RowColumn * (n_ext + n_base))()
    for c in range(n_ext + n_base):
        cols[c].d_values, cols[c].is_ext, cols[c].field_id = None, int(c < n_ext), 0
    hdr = (u32 * 4)()
    steps = np.zeros(2 * 4096, dtype=np.uint64)
    ints = np.zeros(256, dtype=np.uint32)
    h = ctypes.c_uint64()
    _lib.check(lib.bfs_row_template_steps(cols, n_ext + n_base, code, 1, hdr, steps.ctypes.data, 4096, ints.ctypes.data, 256, ctypes.byref(h)))
    header = open(os.path.join(ROOT, "stark_brainfuck_amd", "csrc", "rows_generated.hpp")).read()
    assert "0x%016xull" % h.value in header
    assert [hdr[1], hdr[2], hdr[3]] == [info[1], info[2], info[3]]
