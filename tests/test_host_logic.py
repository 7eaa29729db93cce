"""CPU tests of the product's host side: the C-ABI library loads and exports every declared symbol, the scalar object
model, the native pickle emitter / SHAKE256 transcript (against reference byte strings), index sampling."""
import ctypes
import hashlib
import os
import re

import pytest

from conftest import GOLDEN, ROOT, golden_bytes, load_golden

SEED = 0x5EED


@pytest.fixture(scope="session")
def sb():
    from stark_brainfuck_amd import build
    build.build_library()
    import stark_brainfuck_amd
    return stark_brainfuck_amd


def test_library_exports_every_declared_symbol(sb):
    from stark_brainfuck_amd import _lib
    header = open(os.path.join(ROOT, "include", "bfstark.h")).read()
    declared = set(re.findall(r"\b(bfs_[a-z0-9_]+)\s*\(", header))
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), "libbfstark_hip.so does not export %s" % name
    assert declared == set(_lib.exported_symbols()), declared ^ set(_lib.exported_symbols())
    assert _lib.load().bfs_version() >= 1


def test_route_measurement_report_before_any_measurement(sb):
    """bfs_ntt_route_probe_info (diagnostics of bfs_gl_ntt's route measurement; bench.py prints it): no measurement yet -> zeros, route -1,
    count 0; every pointer may be NULL"""
    from stark_brainfuck_amd import _lib
    lib = _lib.load()
    us, route, probes = (ctypes.c_float * 4)(1, 2, 3, 4), ctypes.c_int(7), ctypes.c_ulonglong(9)
    assert lib.bfs_ntt_route_probe_info(us, ctypes.byref(route), ctypes.byref(probes)) == 0
    import torch
    if not torch.cuda.is_available():
        assert list(us) == [0.0] * 4 and route.value == -1 and probes.value == 0
    assert lib.bfs_ntt_route_probe_info(None, None, None) == 0


def test_no_gpu_means_loud_failure(sb):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    F = sb.BaseField.main()
    with pytest.raises(RuntimeError):
        sb.ntt(F.primitive_nth_root(4), [F(1), F(2), F(3), F(4)])


def test_scalar_model_against_reference(sb):
    g = load_golden("field.json")
    F, XF = sb.BaseField.main(), sb.ExtensionField.main()
    for r in g["base_ops"]:
        a, b = F(r["a"]), F(r["b"])
        assert ((a + b).value, (a - b).value, (a * b).value, (-a).value, (a ^ r["e"]).value) == (r["add"], r["sub"], r["mul"], r["neg"], r["pow"])
        if "inv" in r:
            assert a.inverse().value == r["inv"] and (b / a).value == r["b_div_a"]
    xl = lambda e: [c.value for c in e.polynomial.coefficients]
    for r in g["xfe_ops"]:
        a, b = XF.from_limbs(r["a"]), XF.from_limbs(r["b"])
        assert (xl(a + b), xl(a - b), xl(a * b), xl(-a), xl(a ^ r["e"])) == (r["add"], r["sub"], r["mul"], r["neg"], r["pow"])
        if "inv" in r:
            assert xl(a.inverse()) == r["inv"] and xl(b / a) == r["b_div_a"]
    for r in g["base_sample"]:
        assert F.sample(bytes.fromhex(r["bytes"])).value == r["value"]
    for r in g["xfe_sample"]:
        assert xl(XF.sample(bytes.fromhex(r["bytes"]))) == r["value"]
    for k, v in g["roots"].items():
        assert F.primitive_nth_root(1 << int(k)).value == v
    for k, v in g["xfe_call"].items():
        assert xl(XF(int(k))) == v
    assert F.generator().value == g["generator"]


def test_host_scalar_entry_points(sb):
    from stark_brainfuck_amd import _lib
    lib = _lib.load()
    g = load_golden("field.json")
    for r in g["base_ops"]:
        assert lib.bfs_gl_mul(r["a"], r["b"]) == r["mul"] and lib.bfs_gl_pow(r["a"], r["e"]) == r["pow"]
        if "inv" in r:
            assert lib.bfs_gl_inv(r["a"]) == r["inv"]
    for k, v in g["roots"].items():
        assert lib.bfs_gl_primitive_root(int(k)) == v
    for r in g["xfe_sample"]:
        out = (ctypes.c_uint64 * 3)()
        b = bytes.fromhex(r["bytes"])
        lib.bfs_xfe_sample(b, len(b), out)
        assert [x for x in list(out)][:len(r["value"])] == r["value"]


def test_polynomial_helpers(sb):
    F = sb.BaseField.main()
    P = sb.Polynomial
    a = P([F(3), F(0), F(5), F(7)])
    b = P([F(2), F(1)])
    q, r = P.divide(a, b)
    assert q * b + r == a and r.degree() < b.degree()
    assert (a * b) / b == a
    x, y, g = P.xgcd(a, b)
    assert x * a + y * b == g
    dom = [F(1), F(2), F(5), F(9)]
    vals = a.evaluate_domain(dom)
    assert P.interpolate_domain(dom, vals) == a
    assert P.zerofier_domain(dom).evaluate_domain(dom) == [F(0)] * 4
    assert a.scale(F(3)).evaluate(F(2)) == a.evaluate(F(6))
    assert sb.colinear([(F(1), F(2)), (F(2), F(4)), (F(3), F(6))]) and not sb.colinear([(F(1), F(2)), (F(2), F(4)), (F(3), F(7))])


def test_reference_pickle_of_elements(sb):
    g = load_golden("pickle.json")
    XF, F = sb.ExtensionField.main(), sb.BaseField.main()
    for r in g["xfe_leaves"]:
        assert sb.reference_pickle(XF.from_limbs(r["limbs"])).hex() == r["pickle"]
    for r in g["bfe_leaves"][:8]:
        assert sb.reference_pickle(F(r["limbs"][0])).hex() == r["pickle"]
    # arbitrary picklable leaves go through CPython's own pickle, like the reference
    from stark_brainfuck_amd.merkle import leaf_bytes
    import pickle
    leaf = [b"abc", bytes(range(200))]
    assert leaf_bytes(leaf) == pickle.dumps(leaf, protocol=4)


def test_proof_stream_known_answers(sb):
    g = load_golden("pickle.json")
    XF = sb.ExtensionField.main()
    ps = sb.ProofStream()
    assert ps.serialize() == bytes.fromhex("80045d942e")                 # empty list
    alpha0 = XF.sample(ps.prover_fiat_shamir())
    assert alpha0.limbs() == [12226376460829714024, 16869843892243676816, 17476883614840174052]
    for rec in g["root_lists"]:
        ps = sb.ProofStream()
        for x in rec["roots"]:
            ps.push(bytes.fromhex(x))
        assert ps.serialize().hex() == rec["pickle"] and ps.prover_fiat_shamir().hex() == rec["shake256_32"]
    from oracle.ref_oracle import felt
    r = [hashlib.blake2b(bytes([i])).digest() for i in range(4)]
    e = [XF.from_limbs([felt(SEED + 88, 3 * i + k) for k in range(3)]) for i in range(6)]
    z, salt = XF.zero(), bytes(range(24))
    B = lambda v: sb.BaseFieldElement(v, XF.modulus.coefficients[0].field)
    recipes = {
        "roots_then_codeword": [r[0], r[1], [e[0], e[1], e[2]]],
        "shared_objects": [r[0], [e[0], e[1], e[2], z], (e[0], e[3], e[1]), [r[2], r[3], r[2]]],
        "tuples_first": [(e[4], e[5], e[4]), [r[1]], (z, XF.from_limbs([5]), XF.from_limbs([0, 6]))],
        "with_bfe_and_salt": [B(5), r[0], (B(6), e[0]), [salt, [r[1]]]],
    }
    for rec in g["transcripts"]:
        ps = sb.ProofStream()
        for o in recipes[rec["name"]]:
            ps.push(o)
        assert ps.serialize().hex() == rec["pickle"], rec["name"]
        assert ps.prover_fiat_shamir().hex() == rec["fiat_shamir"]
    ps.read_index = 2
    t2 = sb.ProofStream()
    t2.objects = ps.objects[:2]
    assert ps.verifier_fiat_shamir() == t2.prover_fiat_shamir()


@pytest.mark.parametrize("tag", ["d16_t2", "d64_t8", "d1024_t4", "test_fri_valid", "test_fri_disturbed", "d16_t2_prepushed"])
def test_reference_streams_round_trip(sb, tag):
    """load a proof stream written by the reference into this package's classes and serialise it again with the
    native emitter: must reproduce the reference's bytes (memoisation, opcodes, 64 KiB framing)."""
    blob = golden_bytes("fri_%s_stream.bin" % tag)
    rec = load_golden("fri.json")[tag]
    ps = sb.ProofStream().deserialize(blob)
    assert len(ps.objects) == rec["num_objects"]
    assert ps.serialize() == blob
    assert ps.prover_fiat_shamir().hex() == rec["final_fiat_shamir"]
    xl = lambda el: [c.value for c in el.polynomial.coefficients]
    last = ps.objects[rec["num_prepushed"] + rec["rounds"] - 1]
    assert [xl(el) for el in last] == rec["last_codeword"]


def test_sample_indices(sb):
    XF = sb.ExtensionField.main()
    F = XF.modulus.coefficients[0].field
    fri = sb.Fri(F.generator(), F.primitive_nth_root(64), 64, 4, 2, XF)
    g = load_golden("fri.json")
    for r in g["sample_indices"]:
        assert fri.sample_indices(bytes.fromhex(r["seed"]), r["size"], r["reduced_size"], r["number"]) == r["indices"]
    with pytest.raises(AssertionError, match="cannot sample more indices"):
        fri.sample_indices(b"s", 32, 8, 9)
    assert fri.num_rounds() == 4
    with pytest.raises(AssertionError, match="less than one round"):
        sb.Fri(F.generator(), F.primitive_nth_root(4), 4, 4, 1, XF)


def test_speculative_fiat_shamir_equals_hashing_afterwards(sb):
    """what the FRI prover does while a tree kernel runs (Transcript::speculate / resolve): a placeholder digest is pushed, the SHAKE256
    blocks in front of its payload are absorbed, then the digest is filled in.  Bytes and challenge must equal CPython's
    pickle + hashlib on the finished list -- for streams of any length, across the 64 KiB frame boundary of pickle protocol 4"""
    import ctypes
    import hashlib
    import pickle
    import random
    from stark_brainfuck_amd import _lib
    from stark_brainfuck_amd.ip import NativeTranscript
    lib = _lib.load()
    rng = random.Random(3)
    for prefix in (0, 1, 2, 7, 60, 1100):
        t = NativeTranscript()
        objects = []
        for k in range(prefix):
            obj = bytes(rng.randrange(256) for _ in range(rng.choice((0, 5, 64, 136, 137))))
            objects.append(obj)
            t.push(obj)
        for step in range(40):
            digest = bytes(rng.randrange(256) for _ in range(64))
            out = ctypes.create_string_buffer(32)
            _lib.check(lib.bfs_ps_push_digest_fiat_shamir(t.handle, digest, out, 32))
            objects.append(digest)
            want = pickle.dumps(objects, protocol=4)
            assert t.serialize() == want, (prefix, step)
            assert out.raw == hashlib.shake_256(want).digest(32), (prefix, step)


def test_verifier_shortcuts_answer_like_the_polynomial_routines(sb):
    """Fri.verify asks two polynomial questions (fri.py:253-259, 275-283) and answers them on integer residues: the degree of the
    interpolant of the last codeword over its coset (fri._interpolant_degree) and whether three points lie on a line of degree exactly
    one (fri._on_a_line).  Both against the Lagrange interpolation they replace (Polynomial.interpolate_domain, univariate.colinear's
    general branch), on random and on degenerate inputs."""
    import random
    from stark_brainfuck_amd import fri as fri_module
    from stark_brainfuck_amd.algebra import BaseField, BaseFieldElement
    from stark_brainfuck_amd.extension_field import ExtensionField
    from stark_brainfuck_amd.univariate import Polynomial
    F, XF = BaseField.main(), ExtensionField.main()
    p = F.p
    rng = random.Random(19)

    def xfe():
        return XF.from_limbs([rng.randrange(p) for _ in range(3)])

    for n in (1, 2, 4, 8, 16):
        omega = F.primitive_nth_root(n) if n > 1 else BaseFieldElement(1, F)
        offset = BaseFieldElement(rng.randrange(1, p), F)
        domain = [XF.lift(offset * (omega ^ i)) for i in range(n)]
        for degree in list(range(-1, n)):
            poly = Polynomial([xfe() for _ in range(degree)] + ([xfe()] if degree >= 0 else []))      # leading coefficient non-zero w.h.p.
            if degree >= 0 and poly.degree() != degree:
                continue
            values = poly.evaluate_domain(domain) if degree >= 0 else [XF.zero() for _ in range(n)]
            assert Polynomial.interpolate_domain(domain, values).degree() == degree
            assert fri_module._interpolant_degree(omega.value, [tuple(v.limbs()) for v in values]) == degree, (n, degree)
    for case in range(200):
        ax, bx = rng.randrange(p), rng.randrange(p)
        cx = xfe()
        slope, intercept = (XF.zero() if case % 7 == 0 else xfe()), xfe()
        line = lambda x: slope * x + intercept
        pa, pb, pc = XF.lift(BaseFieldElement(ax, F)), XF.lift(BaseFieldElement(bx, F)), cx
        ya, yb, yc = line(pa), line(pb), line(pc)
        if case % 3 == 0:
            yc = yc + XF.one()                                   # off the line
        want = Polynomial.interpolate_domain([pa, pb, pc], [ya, yb, yc]).degree() == 1
        got = fri_module._on_a_line(ax, tuple(ya.limbs()), bx, tuple(yb.limbs()), tuple(cx.limbs()), tuple(yc.limbs()))
        assert got is want, case
    assert fri_module._on_a_line(5, (1, 0, 0), 5, (2, 0, 0), (9, 1, 0), (3, 0, 0)) is None          # coinciding abscissae
    assert fri_module._on_a_line(5, (1, 0, 0), 6, (2, 0, 0), (5, 0, 0), (3, 0, 0)) is None


def test_native_sampling_equals_the_reference_formulas(sb):
    """bfs_gl_sample = BaseField.sample (algebra.py:138-142: big-endian integer mod p) for every length the callers use and the edge
    values; bfs_sample_weights = BrainfuckStark.sample_weights (brainfuck_stark.py:114-115): ExtensionField.sample of
    blake2b(randomness + bytes(i)), i.e. the three 21-byte chunks of each digest"""
    import ctypes
    import random
    from hashlib import blake2b
    from stark_brainfuck_amd import _lib
    from stark_brainfuck_amd.brainfuck_stark import BrainfuckStark
    lib = _lib.load()
    p = (1 << 64) - (1 << 32) + 1
    rng = random.Random(11)
    for length in range(0, 41):
        for case in range(40):
            data = bytes([255] * length) if case == 0 else bytes(length) if case == 1 else bytes(rng.randrange(256) for _ in range(length))
            assert lib.bfs_gl_sample(data, length) == int.from_bytes(data, "big") % p, (length, data)
    top = p.to_bytes(8, "big")
    for data in (top, top * 2, top * 3, b"\x01" + top, (p - 1).to_bytes(8, "big") * 2 + b"\xff"):
        assert lib.bfs_gl_sample(data, len(data)) == int.from_bytes(data, "big") % p
    for number, length in ((0, 32), (1, 32), (4, 32), (5, 32), (157, 32), (300, 0), (11, 64), (200, 131)):
        randomness = bytes(rng.randrange(256) for _ in range(length))
        want = []
        for i in range(number):
            digest = blake2b(randomness + bytes(i)).digest()
            want.append(tuple(int.from_bytes(digest[21 * k:21 * k + 21], "big") % p for k in range(3)))
        assert BrainfuckStark._sample_weights(number, randomness) == want, (number, length)
        raw = (ctypes.c_uint64 * max(3 * number, 1))()
        _lib.check(lib.bfs_sample_weights(randomness, length, number, raw))
        assert [tuple(raw[3 * i:3 * i + 3]) for i in range(number)] == want


def test_lookahead_fiat_shamir_equals_hashing_afterwards(sb):
    """what the FRI prover does behind a long transcript (Transcript::Lookahead): the pickles of the stream plus 1, 2, ... count
    placeholder digests are made up front, helper threads absorb each one's SHAKE256 blocks in front of the first placeholder, the real
    digests are filled in as they arrive.  Bytes and challenges must equal CPython's pickle + hashlib on the growing list -- also when
    the run crosses the 64 KiB frame boundary or the 1000-item batch boundary of pickle protocol 4, and the stream must be usable as
    before afterwards."""
    import ctypes
    import hashlib
    import pickle
    import random
    from stark_brainfuck_amd import _lib
    from stark_brainfuck_amd.ip import NativeTranscript
    lib = _lib.load()
    rng = random.Random(5)
    took_the_lookahead_route = 0
    # (466 objects of 137 bytes are 65 240 bytes of pickle: the fifth digest of the run takes the frame over 64 KiB, so the frame that is
    # open when the run starts is closed in the middle of it)
    for prefix, count, size in ((0, 3, None), (1, 3, None), (2, 5, None), (7, 20, None), (60, 14, None), (990, 25, None), (1100, 22, None),
                                (470, 40, None), (466, 12, 137), (465, 9, 137), (467, 30, 137)):
        t = NativeTranscript()
        objects = []
        for k in range(prefix):
            obj = bytes(rng.randrange(256) for _ in range(size if size is not None else rng.choice((0, 5, 64, 136, 137))))
            objects.append(obj)
            t.push(obj)
        for run in range(2):              # a second run on the same stream: the tentative items of the first left no trace
            digests = bytes(rng.randrange(256) for _ in range(64 * count))
            out = ctypes.create_string_buffer(32 * count)
            used = ctypes.c_int(-1)
            _lib.check(lib.bfs_ps_push_digests_fiat_shamir(t.handle, digests, count, out, 32, ctypes.byref(used)))
            took_the_lookahead_route += used.value
            for k in range(count):
                objects.append(digests[64 * k:64 * k + 64])
                want = pickle.dumps(objects, protocol=4)
                assert out.raw[32 * k:32 * k + 32] == hashlib.shake_256(want).digest(32), (prefix, run, k)
            assert t.serialize() == pickle.dumps(objects, protocol=4), (prefix, run)
            extra = bytes(rng.randrange(256) for _ in range(33))
            objects.append(extra)
            t.push(extra)
            assert t.serialize() == pickle.dumps(objects, protocol=4), (prefix, run)
    assert took_the_lookahead_route >= 10          # (streams of fewer than two objects fall back)


def test_helper_threads_are_restarted_in_a_forked_child(sb):
    """the helper pool (csrc/helper_pool.hpp) belongs to the process that started it: a fork()ed child has the pool object but none
    of its threads, and must start its own instead of waiting for jobs nobody runs"""
    import ctypes
    import hashlib
    import os
    import pickle
    from stark_brainfuck_amd import _lib
    from stark_brainfuck_amd.ip import NativeTranscript
    lib = _lib.load()

    def run(tag):
        t = NativeTranscript()
        objects = [bytes([k]) * 40 for k in range(200)]
        for obj in objects:
            t.push(obj)
        digests = hashlib.shake_256(tag).digest(64 * 6)
        out = ctypes.create_string_buffer(32 * 6)
        used = ctypes.c_int(-1)
        _lib.check(lib.bfs_ps_push_digests_fiat_shamir(t.handle, digests, 6, out, 32, ctypes.byref(used)))
        for k in range(6):
            objects.append(digests[64 * k:64 * k + 64])
            if out.raw[32 * k:32 * k + 32] != hashlib.shake_256(pickle.dumps(objects, protocol=4)).digest(32):
                return False
        return used.value == 1

    assert run(b"parent")
    pid = os.fork()
    if pid == 0:
        code = 3
        try:
            code = 0 if run(b"child") else 1
        finally:
            os._exit(code)
    _, status = os.waitpid(pid, 0)
    assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0, status
    assert run(b"parent again")


def test_fastlist_conversions_match_the_python_loops():
    """cpyext/fastlist.c: lists of element objects <-> uint64 buffers.  Same values, same object structure (trimmed coefficient
    lists, field references) -- checked through CPython's pickle of both results -- and the fallbacks for odd inputs."""
    import pickle
    import time
    import numpy as np
    from stark_brainfuck_amd import arrays
    from stark_brainfuck_amd.algebra import BaseField, BaseFieldElement
    from stark_brainfuck_amd.extension_field import ExtensionField, ExtensionFieldElement
    from stark_brainfuck_amd.univariate import Polynomial
    if arrays._fastlist is None:             # a tree where build() has not run yet: build the helper now (gcc, a second)
        import importlib
        from stark_brainfuck_amd import build
        build.build_fastlist()
        importlib.reload(arrays)
    fl = arrays._fastlist
    assert fl is not None, "the helper is built by __graft_entry__.build()"
    P = (1 << 64) - (1 << 32) + 1
    rng = np.random.default_rng(5)
    n = 5000
    vals = rng.integers(0, P, n, dtype=np.uint64)
    vals[:6] = [0, 1, 255, 1 << 31, 1 << 63, P - 1]
    F = BaseField.main()
    elems = [BaseFieldElement(int(v), F) for v in vals]
    out = np.empty(n, dtype=np.uint64)
    assert fl.pack_base(elems, out) == n and (out == vals).all()
    assert fl.pack_base(tuple(elems[:7]), out) == 7                       # any sequence
    back = fl.unpack_base(vals, BaseFieldElement, F)
    assert [e.value for e in back] == [int(v) for v in vals] and all(e.field is F for e in back) and type(back[0]) is BaseFieldElement
    assert pickle.dumps(back) == pickle.dumps(elems)
    XF = ExtensionField.main()
    soa = rng.integers(0, P, (3, n), dtype=np.uint64)
    soa[:, 0] = 0                               # the zero element: no coefficients
    soa[1:, 1] = 0                              # one coefficient
    soa[2, 2] = 0                               # two
    soa[0, 3] = 0                               # a zero in front stays (only trailing zeros are trimmed)
    xs = [XF.from_limbs([int(soa[0, i]), int(soa[1, i]), int(soa[2, i])]) for i in range(n)]
    got = fl.unpack_ext(np.ascontiguousarray(soa), ExtensionFieldElement, Polynomial, BaseFieldElement, XF, XF._base())
    assert [len(e.polynomial.coefficients) for e in got[:4]] == [0, 1, 2, 3]
    assert pickle.dumps(got) == pickle.dumps(xs)
    out3 = np.empty((3, n), dtype=np.uint64)
    assert fl.pack_ext(xs, out3) == n and (out3 == soa).all()
    # out-of-range values go to the Python path
    with pytest.raises(OverflowError):
        fl.pack_base([BaseFieldElement(1 << 64, F)], out)
    with pytest.raises(OverflowError):
        fl.pack_base([BaseFieldElement(-1, F)], out)
    with pytest.raises(AttributeError):
        fl.pack_base([object()], out)
    with pytest.raises(ValueError):
        fl.pack_base(elems, np.empty(3, dtype=np.uint64))
    # and it is the point of the exercise: an order of magnitude faster than the loops (measured: base 170 ns vs 2.2 us per element
    # out, 60 vs 150 ns in; extension 1.1 vs 26 us out, 0.3 vs 1.4 us in)
    # (best of three with the collector off: in the middle of a full test run an allocation-triggered collection over the objects
    #  of earlier tests lands in either measurement; the helper only has to be clearly faster, the ratios are in DESIGN.md)
    import gc
    big = np.tile(vals, 20)
    gc.collect()
    gc.disable()
    try:
        t_c, t_py = [], []
        for _ in range(3):
            t0 = time.perf_counter(); keep = fl.unpack_base(big, BaseFieldElement, F); t_c.append(time.perf_counter() - t0)
            del keep
            t0 = time.perf_counter(); keep = [BaseFieldElement(int(v), F) for v in big]; t_py.append(time.perf_counter() - t0)
            del keep
    finally:
        gc.enable()
    assert min(t_c) * 1.5 < min(t_py), (t_c, t_py)


def test_randomness_override_is_per_context_not_process_global():
    """shard.shared_randomness: another prover thread inside the context keeps its own random source, and the
    module-level `urandom` names -- which the reference's tests patch -- are not touched"""
    import os
    import threading
    from stark_brainfuck_amd import brainfuck_stark, randomness, salted_merkle, shard, table
    seen = {}
    with shard.shared_randomness(1, 0, seed=b"s" * 32) as stream:
        assert randomness.source(os.urandom) is stream
        assert brainfuck_stark.urandom is os.urandom and table.urandom is os.urandom and salted_merkle.urandom is os.urandom
        t = threading.Thread(target=lambda: seen.update(other=randomness.source(os.urandom)))
        t.start(); t.join()
        first = stream(16)
    assert seen["other"] is os.urandom
    assert randomness.source(os.urandom) is os.urandom
    with shard.shared_randomness(1, 0, seed=b"s" * 32) as again:
        assert again(16) == first and again(8) != first[:8]          # deterministic in the seed, and a stream, not a function of the count


def test_row_window_entry_points_reject_windows_outside_the_domain():
    """bfs_*_rows (the share of one rank of a cooperative proof): a window that does not lie inside the 2^log_n points is refused with
    BFS_ERR_BAD_ARG before anything touches the device (so this runs without a GPU), and an empty window is a no-op"""
    import ctypes
    from stark_brainfuck_amd import _lib
    lib = _lib.load()
    u64 = ctypes.c_uint64
    one = (ctypes.c_uint32 * 1)(0)
    val = (u64 * 1)(1)
    for first, count in ((17, 0), (16, 1), (0, 17), (8, 9), (1 << 63, 1 << 63)):
        rc = lib.bfs_zerofier_inverses_rows(4, 7, 1, 1, one, val, None, first, count, None)
        assert rc == 6, (first, count, rc)
        assert b"not inside the domain" in lib.bfs_last_error()
    w = _lib.CombWeight()
    assert lib.bfs_difference_combine_rows(None, None, 4, 7, 1, ctypes.byref(w), None, None, 12, 5, None) == 6
    assert lib.bfs_air_combine_rows(0, None, None, 4, 1, 0, 1, 7, 1, None, None, None, None, None, None, None, None, 3, 14, None) == 6
    # an empty window inside the domain: nothing to do, no device needed
    assert lib.bfs_zerofier_inverses_rows(4, 7, 1, 1, one, val, None, 16, 0, None) == 0
    assert lib.bfs_difference_combine_rows(None, None, 4, 7, 1, ctypes.byref(w), None, None, 5, 0, None) == 0


def test_shipped_library_has_no_experiment_switches_and_no_carry_hazards():
    """round-3 verdict #6 / advice: (1) no timing-only (wrong-result) or A/B branch is left in the product sources and the build
    defines none; (2) the library in the tree was built from exactly these sources with exactly the default flags; (3) the gfx950
    listings of that build hold no VALU carry read closer than two wait states to the VALU write of its SGPR pair
    (tools/isa_hazards.py: the padding inside inline asm is ours to get right)."""
    import subprocess
    import sys
    from stark_brainfuck_amd import build as b
    hits = []
    for f in sorted(os.listdir(b.CSRC)):
        text = open(os.path.join(b.CSRC, f), errors="replace").read()
        hits += ["%s: %s" % (f, m) for m in re.findall(r"BFS_\w*ABL\w*", text)]
    assert not hits, hits
    assert not [f for f in b.FLAGS if f.startswith("-DBFS_")]
    b.build_library()                                   # no-op when current; rebuilds (with the default flags) otherwise
    assert b.up_to_date(), "libbfstark_hip.so is not what the default flags build from the sources in the tree"
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_hazards.py")], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-1000:]
    assert " 0 site(s)" in res.stdout
