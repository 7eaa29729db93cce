"""Pins the CPU oracle (oracle/) against golden vectors captured from the reference itself
(tests/golden/*.json, generator: tests/golden/gen_golden.py).  CPU only."""
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden_bytes, load_golden

SEED = 0x5EED


def sha_u64(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype="<u8").tobytes()).hexdigest()


def sha_soa(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype="<u8").tobytes()).hexdigest()


def test_felt_and_roots(oracle):
    g = load_golden("field.json")
    assert [oracle.felt(SEED, i) for i in range(8)] == g["felt"]
    for k, v in g["roots"].items():
        assert oracle.primitive_nth_root(1 << int(k)) == v
    assert oracle.GENERATOR == g["generator"] and oracle.P == g["p"]


def test_base_ops(oracle):
    for r in load_golden("field.json")["base_ops"]:
        a, b = r["a"], r["b"]
        assert oracle.add(a, b) == r["add"] and oracle.sub(a, b) == r["sub"] and oracle.mul(a, b) == r["mul"]
        assert oracle.neg(a) == r["neg"] and oracle.power(a, r["e"]) == r["pow"]
        if "inv" in r:
            assert oracle.inv(a) == r["inv"] and oracle.mul(b, oracle.inv(a)) == r["b_div_a"]
    for r in load_golden("field.json")["base_sample"]:
        assert oracle.sample(bytes.fromhex(r["bytes"])) == r["value"]


def test_xfe_ops(oracle):
    g = load_golden("field.json")
    t = oracle.xtrim
    for r in g["xfe_ops"]:
        a, b = r["a"], r["b"]
        assert t(oracle.xadd(a, b)) == r["add"] and t(oracle.xsub(a, b)) == r["sub"] and t(oracle.xmul(a, b)) == r["mul"]
        assert t(oracle.xsub([], a)) == r["neg"] and t(oracle.xpow(a, r["e"])) == r["pow"]
        if "inv" in r:
            assert t(oracle.xinv(a)) == r["inv"] and t(oracle.xmul(b, oracle.xinv(a))) == r["b_div_a"]
    for r in g["xfe_sample"]:
        assert t(oracle.xsample(bytes.fromhex(r["bytes"]))) == r["value"]


@pytest.mark.parametrize("logn", [0, 1, 2, 3, 5, 8, 10, 12, 14, 16])
def test_ntt_intt(oracle, logn):
    c = load_golden("ntt.json")["cases"][str(logn)]
    n = 1 << logn
    v = oracle.felt_array(SEED, 0, n)
    assert sha_u64(v) == c["sha_in"]
    fw = oracle.ntt(c["root"], v)
    iv = oracle.intt(c["root"], v)
    assert sha_u64(fw) == c["sha_ntt"] and sha_u64(iv) == c["sha_intt"]
    assert fw[:4].tolist() == c["ntt_head"][:n] and iv[-4:].tolist() == c["intt_tail"][-n:]
    if "ntt" in c:
        assert fw.tolist() == c["ntt"] and iv.tolist() == c["intt"]
    assert oracle.ntt(c["root"], iv).tolist() == v.tolist()


def test_ntt20_if_present(oracle):
    p = os.path.join(GOLDEN, "ntt20.json")
    if not os.path.exists(p):
        pytest.skip("ntt20.json not generated yet")
    c = load_golden("ntt20.json")
    v = oracle.felt_array(SEED, 0, 1 << 20)
    assert sha_u64(v) == c["sha_in"]
    assert sha_u64(oracle.ntt(c["root"], v)) == c["sha_ntt"]
    assert sha_u64(oracle.intt(c["root"], v)) == c["sha_intt"]


def test_pure_python_ntt_restatement(oracle):
    """bench.py's `cpu_baseline.python` leg: the boxed-element restatement of ntt.py:4-23 gives the reference's outputs (goldens
    2^0..2^10 produced by the reference itself) and the C oracle's at 2^11"""
    for logn in range(0, 11):
        c = load_golden("ntt.json")["cases"][str(logn)]
        v = oracle.felt_array(SEED, 0, 1 << logn)
        got = np.array(oracle.ntt_python(c["root"], v.tolist()), dtype=np.uint64)
        assert sha_u64(got) == c["sha_ntt"], logn
    w = oracle.primitive_nth_root(2048)
    v = oracle.felt_array(SEED + 1, 0, 2048)
    assert oracle.ntt_python(w, v.tolist()) == oracle.ntt(w, v).tolist()
    with pytest.raises(AssertionError):
        oracle.ntt_python(3, [1] * 8)


def test_ntt_errors(oracle):
    w8 = oracle.primitive_nth_root(8)
    with pytest.raises(AssertionError, match="non-power-of-two"):
        oracle.ntt(w8, [1] * 6)
    with pytest.raises(AssertionError, match="nth root of unity"):
        oracle.ntt(3, [1] * 8)
    with pytest.raises(AssertionError, match="not primitive"):
        oracle.ntt(oracle.primitive_nth_root(4), [1] * 8)


def test_xfe_ntt(oracle):
    g = load_golden("ntt.json")["xfe_ntt_64"]
    soa = np.array(g["in"], dtype=np.uint64).T
    w = oracle.primitive_nth_root(64)
    assert oracle.xntt_soa(w, soa).T.tolist() == g["ntt"]
    assert oracle.xintt_soa(w, soa).T.tolist() == g["intt"]


def test_fast_multiply_and_coset(oracle):
    g = load_golden("poly.json")
    w = oracle.primitive_nth_root(64)
    for c in g["fast_multiply_n64"]:
        assert oracle.fast_multiply(c["lhs"], c["rhs"], w, 64) == c["product"]
    for c in g["coset"]:
        gen = oracle.primitive_nth_root(c["order"])
        vals = oracle.fast_coset_evaluate(c["coefficients"], c["offset"], gen, c["order"])
        assert vals.tolist() == c["values"]
        assert oracle.fast_coset_interpolate(c["offset"], gen, vals).tolist() == c["interpolated"]
    b = g["batch_inverse"]
    assert oracle.batch_inverse(b["in"]).tolist() == b["out"]
    with pytest.raises(AssertionError, match="zero"):
        oracle.batch_inverse([1, 0])
    d = g["domain64"]
    assert oracle.fast_coset_evaluate(d["coefficients"], d["offset"], d["omega"], 64).tolist() == d["evaluate"]
    xs = np.array(d["xcoefficients"], dtype=np.uint64).T
    assert oracle.xevaluate_soa(xs, d["offset"], d["omega"], 64).T.tolist() == d["xevaluate"]


def test_fast_family_over_the_extension_field(oracle):
    """ntt.py:45-79, 177-235 on ExtensionFieldElement operands with a lifted root -- the call Table.ldex makes (table.py:133-134)"""
    g = load_golden("polyx.json")
    for c in g["fast_multiply"]:
        w = oracle.primitive_nth_root(c["order"])
        assert oracle.xfast_multiply(c["lhs"], c["rhs"], w, c["order"]) == c["product"]
        if "quotient_by_lhs" in c and oracle._xdegree(c["product"]) >= 8:
            assert oracle.xfast_coset_divide(c["product"], c["lhs"], 7, w, c["order"]) == c["quotient_by_lhs"]
    b = g["batch_inverse"]
    assert oracle._xlist(oracle.xbatch_inverse(oracle._xsoa(b["in"]))) == b["out"]
    assert all(oracle.xmul(x, y) == [1, 0, 0] for x, y in zip(b["in"], b["out"]))
    with pytest.raises(AssertionError, match="zero"):
        oracle.xbatch_inverse(oracle._xsoa([[1, 2, 3], [0, 0, 0]]))
    # the interpolants the reference's subproduct tree returns take the given values on the domain of table.py:120-124
    for c in g["interpolate_columns"]:
        for x, v in zip(c["domain"], c["values"]):
            acc = [0, 0, 0]
            for coeff in reversed(c["interpolant"]):
                acc = oracle.xadd(oracle.xmul(acc, x), coeff)
            assert acc == v


def test_leaf_pickles(oracle):
    g = load_golden("pickle.json")
    for r in g["bfe_leaves"]:
        bs = oracle.dumps(oracle.make_bfe(r["limbs"][0]))
        assert bs.hex() == r["pickle"] and hashlib.blake2b(bs).hexdigest() == r["blake2b"]
    for r in g["xfe_leaves"]:
        assert oracle.dumps(oracle.make_xfe(r["limbs"])).hex() == r["pickle"]
    assert oracle.dumps(bytes.fromhex(g["salt"]["salt"])).hex() == g["salt"]["pickle"]


def test_transcripts(oracle):
    g = load_golden("pickle.json")
    r = [hashlib.blake2b(bytes([i])).digest() for i in range(4)]
    for rec in g["root_lists"]:
        ps = oracle.ProofStreamOracle()
        for x in rec["roots"]:
            ps.push(bytes.fromhex(x))
        assert ps.serialize().hex() == rec["pickle"] and ps.prover_fiat_shamir().hex() == rec["shake256_32"]
    e = [oracle.make_xfe([oracle.felt(SEED + 88, 3 * i + k) for k in range(3)]) for i in range(6)]
    z = oracle.make_xfe([])
    salt = bytes(range(24))
    B = lambda v: oracle.make_bfe(v, internal=True)
    recipes = {
        "roots_then_codeword": [r[0], r[1], [e[0], e[1], e[2]]],
        "shared_objects": [r[0], [e[0], e[1], e[2], z], (e[0], e[3], e[1]), [r[2], r[3], r[2]]],
        "tuples_first": [(e[4], e[5], e[4]), [r[1]], (z, oracle.make_xfe([5]), oracle.make_xfe([0, 6]))],
        "with_bfe_and_salt": [B(5), r[0], (B(6), e[0]), [salt, [r[1]]]],
    }
    for rec in g["transcripts"]:
        ps = oracle.ProofStreamOracle()
        for o in recipes[rec["name"]]:
            ps.push(o)
        assert ps.serialize().hex() == rec["pickle"], rec["name"]
        assert ps.prover_fiat_shamir().hex() == rec["fiat_shamir"]


def test_merkle(oracle):
    g = load_golden("merkle.json")
    for t in g["xfe_trees"]:
        n = t["n"]
        leaves = [oracle.make_xfe([oracle.felt(SEED + t["seed_offset"], 3 * i + k) for k in range(3)]) for i in range(n)]
        m = oracle.MerkleOracle([oracle.dumps(l) for l in leaves])
        assert m.depth == t["depth"] and m.root().hex() == t["root"]
        assert [x.hex() for x in m.nodes] == t["nodes"]
        for i in range(n):
            assert [x.hex() for x in m.open(i)] == t["paths"][i]
            assert oracle.merkle_verify(m.root(), i, m.open(i), oracle.dumps(leaves[i]))
    t = g["bfe_tree"]
    m = oracle.MerkleOracle([oracle.dumps(oracle.make_bfe(oracle.felt(SEED + t["seed_offset"], i))) for i in range(t["n"])])
    assert m.root().hex() == t["root"]
    s = g["salted_tree"]
    salts = [hashlib.shake_256(b"salt" + (i + 1).to_bytes(8, "little")).digest(24) for i in range(s["n"])]
    assert [x.hex() for x in salts] == s["salts"]
    leaves = [oracle.make_xfe([oracle.felt(SEED + s["seed_offset"], 3 * i + k) for k in range(3)]) for i in range(s["n"])]
    m = oracle.MerkleOracle([oracle.salted_leaf_bytes(l, sa) for l, sa in zip(leaves, salts)])
    assert m.root().hex() == s["root"] and [x.hex() for x in m.nodes] == s["nodes"]


def test_sample_indices(oracle):
    for r in load_golden("fri.json")["sample_indices"]:
        assert oracle.sample_indices(bytes.fromhex(r["seed"]), r["size"], r["reduced_size"], r["number"]) == r["indices"]


def _codeword(oracle, rec, coeff):
    d = 1 << rec["log_degree"]
    cw = oracle.xevaluate_soa(coeff(d), rec["offset"], rec["omega"], rec["N"])
    for i in rec.get("disturb", []):
        cw[:, i] = 0
    assert sha_soa(cw) == rec["codeword_sha"]
    return cw


def _seeded(oracle):
    return lambda d: oracle.felt_array(SEED, 0, 3 * d).reshape(d, 3).T.copy()


def _counting(d):
    a = np.zeros((3, d), dtype=np.uint64)
    a[0] = np.arange(d, dtype=np.uint64)
    return a


@pytest.mark.parametrize("tag", ["d16_t2", "d64_t8", "d1024_t4", "test_fri_valid", "test_fri_disturbed", "d16_t2_prepushed"])
def test_fri_prove(oracle, tag):
    rec = load_golden("fri.json")[tag]
    cw = _codeword(oracle, rec, _counting if tag.startswith("test_fri") else _seeded(oracle))
    ps = oracle.ProofStreamOracle()
    if rec["num_prepushed"]:
        r = [hashlib.blake2b(bytes([i])).digest() for i in range(2)]
        e = [oracle.make_xfe([oracle.felt(SEED + 88, 3 * i + k) for k in range(3)]) for i in range(3)]
        for o in [r[0], (e[0], e[1], e[2]), [r[1]]]:
            ps.push(o)
    out = oracle.fri_prove(cw, rec["offset"], rec["omega"], rec["expansion"], rec["num_colinearity_tests"], ps)
    assert [x.hex() for x in out["roots"]] == rec["roots"]
    assert out["alphas"] == rec["alphas"]
    assert [sha_soa(c) for c in out["codewords"]] == rec["codeword_shas"]
    assert [oracle.xtrim(out["last_codeword"][:, i]) for i in range(out["last_codeword"].shape[1])] == rec["last_codeword"]
    assert out["indices"] == rec["indices"]
    assert len(ps.objects) == rec["num_objects"]
    ser = ps.serialize()
    assert hashlib.sha256(ser).hexdigest() == rec["serialize_sha256"] and len(ser) == rec["serialize_len"]
    assert ser == golden_bytes("fri_%s_stream.bin" % tag)
    assert ps.prover_fiat_shamir().hex() == rec["final_fiat_shamir"]
