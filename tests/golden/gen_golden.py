#!/usr/bin/env python3
"""Golden-vector generator: imports the *reference* (aszepieniec/stark-brainfuck, pure Python)
and records inputs/outputs of its polynomial hot path as small data fixtures.

Runs ONLY in the build container (needs /root/reference); the fixtures it writes under
tests/golden/ are what travels to the GPU box.  Nothing of the reference's source is copied:
the fixtures are numbers, digests and pickle byte strings produced by running it.

    python tests/golden/gen_golden.py small     # seconds..minutes: field, ntt<=2^16, pickles, merkle, fri<=4096
    python tests/golden/gen_golden.py ntt20     # ~8 min : 2^20 base NTT + INTT   (BASELINE config 2)
    python tests/golden/gen_golden.py fri20     # ~1.5 h : FRI d=2^18, N=2^20     (BASELINE config 3)

Input recipe: felt(seed, i) = splitmix64(seed + i) mod p, seed 0x5EED.
Extension elements are built "variant A": every coefficient references the BaseField instance that
lives inside the ExtensionField's modulus, which is what the reference's own arithmetic produces
(extension_field.py:88-98, algebra.py:20-30) and what fixes the pickle byte stream.
"""
import sys
sys.dont_write_bytecode = True
import os, json, hashlib, struct, time, pickle

REF = os.environ.get("BFS_REFERENCE", "/root/reference/code")
sys.path.insert(0, REF)
sys.setrecursionlimit(10000)

from algebra import BaseField, BaseFieldElement          # noqa: E402
from univariate import Polynomial                        # noqa: E402
from extension_field import ExtensionField, ExtensionFieldElement  # noqa: E402
import ntt as refntt                                     # noqa: E402
from merkle import Merkle                                # noqa: E402
import salted_merkle                                     # noqa: E402
from salted_merkle import SaltedMerkle                   # noqa: E402
from ip import ProofStream                               # noqa: E402
from fri import Fri                                      # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
P = 18446744069414584321
SEED = 0x5EED
M64 = (1 << 64) - 1


def splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & M64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def felt(seed, i):
    return splitmix64((seed + i) & M64) % P


XF = ExtensionField.main()
BF = XF.modulus.coefficients[0].field   # the internal BaseField instance (variant A)
BF2 = BaseField.main()                  # a separate instance, for pure base-field work


def B(v, field=BF2):
    return BaseFieldElement(v % P, field)


def X(limbs):
    """variant-A extension element from a list of 0..3 limbs (trailing zeros are trimmed by the ctor)."""
    return ExtensionFieldElement(Polynomial([BaseFieldElement(v % P, BF) for v in limbs]), XF)


def xl(e):
    """stored coefficient list of an extension element (trimmed form)."""
    return [c.value for c in e.polynomial.coefficients]


def xl3(e):
    c = xl(e)
    return c + [0] * (3 - len(c))


def sha_u64(vals):
    h = hashlib.sha256()
    h.update(struct.pack("<%dQ" % len(vals), *vals))
    return h.hexdigest()


def sha_xfe_soa(elems):
    """sha256 of c0[0..n) || c1[0..n) || c2[0..n) as little-endian u64 (the SoA device layout)."""
    l3 = [xl3(e) for e in elems]
    h = hashlib.sha256()
    for k in range(3):
        h.update(struct.pack("<%dQ" % len(l3), *[t[k] for t in l3]))
    return h.hexdigest()


def dump(name, obj):
    path = os.path.join(HERE, name)
    with open(path, "w") as f:
        json.dump(obj, f, indent=None, separators=(",", ":"), sort_keys=True)
        f.write("\n")
    print("wrote", path, os.path.getsize(path), "bytes")


def assertion_message(fn):
    try:
        fn()
    except AssertionError as e:
        return str(e)
    return None


# ----------------------------------------------------------------------------------------------
def gen_field():
    out = {"p": P, "seed": SEED, "pickle_protocol": pickle.DEFAULT_PROTOCOL,
           "python": sys.version.split()[0]}
    out["felt"] = [felt(SEED, i) for i in range(8)]
    out["generator"] = BF2.generator().value
    out["roots"] = {str(k): BF2.primitive_nth_root(1 << k).value for k in range(0, 33)}
    base = []
    for t in range(24):
        a, b = felt(SEED + 1000, 2 * t), felt(SEED + 1000, 2 * t + 1)
        if t == 0:
            a, b = 0, 5
        if t == 1:
            a, b = P - 1, P - 1
        if t == 2:
            a, b = (1 << 32), (1 << 32) - 1
        if t == 3:
            a, b = 0xFFFFFFFF00000000, 0xFFFFFFFF
        ea, eb = B(a), B(b)
        e = felt(SEED + 2000, t) >> (t * 2)
        rec = {"a": a, "b": b, "add": (ea + eb).value, "sub": (ea - eb).value, "mul": (ea * eb).value,
               "neg": (-ea).value, "e": e, "pow": (ea ^ e).value}
        if a != 0:
            rec["inv"] = ea.inverse().value
            rec["b_div_a"] = (eb / ea).value
        base.append(rec)
    out["base_ops"] = base
    out["base_sample"] = []
    for ln in (1, 8, 9, 10, 17, 21, 32):
        bs = bytes((splitmix64(7000 + ln * 64 + i) & 0xFF) for i in range(ln))
        out["base_sample"].append({"bytes": bs.hex(), "value": BF2.sample(bs).value})
    xs = []
    for t in range(24):
        la = [felt(SEED + 3000, 6 * t + i) for i in range(3)]
        lb = [felt(SEED + 3000, 6 * t + 3 + i) for i in range(3)]
        if t == 0:
            la = []
        if t == 1:
            la, lb = [5], [0, 0, 9]
        if t == 2:
            la, lb = [0, 7], [3, 4]
        if t == 3:
            la, lb = [P - 1, P - 1, P - 1], [P - 1, P - 1, P - 1]
        if t == 4:
            lb = la
        ea, eb = X(la), X(lb)
        e = felt(SEED + 4000, t) >> (t * 2 + 8)
        rec = {"a": xl(ea), "b": xl(eb), "add": xl(ea + eb), "sub": xl(ea - eb), "mul": xl(ea * eb),
               "neg": xl(-ea), "e": e, "pow": xl(ea ^ e)}
        if not ea.is_zero():
            rec["inv"] = xl(ea.inverse())
            rec["b_div_a"] = xl(eb / ea)
        xs.append(rec)
    out["xfe_ops"] = xs
    out["xfe_sample"] = []
    for ln in (24, 27, 32, 64):
        bs = bytes((splitmix64(9000 + ln * 64 + i) & 0xFF) for i in range(ln))
        out["xfe_sample"].append({"bytes": bs.hex(), "value": xl(XF.sample(bs))})
    out["xfe_call"] = {str(v): xl(XF(v)) for v in (0, 1, 7, 255, 65536)}
    out["xfe_lift"] = xl(XF.lift(B(12345)))
    dump("field.json", out)


def gen_ntt():
    out = {"seed": SEED, "cases": {}}
    for logn in list(range(0, 11)) + [12, 14, 16]:
        n = 1 << logn
        w = BF2.primitive_nth_root(n)
        vals = [felt(SEED, i) for i in range(n)]
        t0 = time.perf_counter()
        fw = [e.value for e in refntt.ntt(w, [B(v) for v in vals])]
        t1 = time.perf_counter()
        iv = [e.value for e in refntt.intt(w, [B(v) for v in vals])] if n > 1 else vals
        t2 = time.perf_counter()
        rec = {"root": w.value, "sha_in": sha_u64(vals), "sha_ntt": sha_u64(fw), "sha_intt": sha_u64(iv),
               "ntt_head": fw[:4], "ntt_tail": fw[-4:], "intt_head": iv[:4], "intt_tail": iv[-4:],
               "ref_seconds": [t1 - t0, t2 - t1]}
        if logn <= 10:
            rec["ntt"] = fw
            rec["intt"] = iv
        out["cases"][str(logn)] = rec
        print("ntt logn", logn, "%.2fs %.2fs" % (t1 - t0, t2 - t1), flush=True)
    # extension-field NTT = three limb NTTs with a lifted root (fri.py:32-37, ntt.py:164-168)
    n = 64
    w = XF.lift(BF.primitive_nth_root(n))
    xv = [X([felt(SEED + 50, 3 * i + k) for k in range(3)]) for i in range(n)]
    out["xfe_ntt_64"] = {"in": [xl3(e) for e in xv], "ntt": [xl3(e) for e in refntt.ntt(w, xv)],
                         "intt": [xl3(e) for e in refntt.intt(w, xv)]}
    # error behaviour (assert messages, ntt.py:5-6,13-16,28-31)
    w8 = BF2.primitive_nth_root(8)
    out["errors"] = {
        "non_pow2": assertion_message(lambda: refntt.ntt(w8, [B(1)] * 6)),
        "not_root": assertion_message(lambda: refntt.ntt(B(3), [B(1)] * 8)),
        "not_primitive": assertion_message(lambda: refntt.ntt(BF2.primitive_nth_root(4), [B(1)] * 8)),
        "intt_non_pow2": assertion_message(lambda: refntt.intt(w8, [B(1)] * 6)),
        "intt_not_root": assertion_message(lambda: refntt.intt(B(3), [B(1)] * 8)),
    }
    dump("ntt.json", out)


def gen_poly():
    out = {}
    n = 64
    w = BF2.primitive_nth_root(n)
    cases = []
    for (dl, dr) in [(-1, 5), (0, 0), (2, 3), (3, 4), (4, 4), (10, 20), (31, 31), (31, 0), (17, 9), (30, 33)]:
        lc = [felt(SEED + 100 + dl, i) for i in range(dl + 1)]
        rc = [felt(SEED + 200 + dr, i) for i in range(dr + 1)]
        if (dl, dr) == (17, 9):
            lc = lc + [0, 0]        # trailing zeros must not change the product
        lhs, rhs = Polynomial([B(v) for v in lc]), Polynomial([B(v) for v in rc])
        prod = refntt.fast_multiply(lhs, rhs, w, n)
        rec = {"lhs": lc, "rhs": rc, "product": [c.value for c in prod.coefficients]}
        if dl >= 0 and dr >= 0:
            q = refntt.fast_coset_divide(prod, lhs, BF2.generator(), w, n)
            rec["quotient_by_lhs"] = [c.value for c in q.coefficients]
        cases.append(rec)
    out["fast_multiply_n64"] = cases
    # coset evaluation / interpolation (ntt.py:164-174)
    ce = []
    for (order, deg, offset) in [(64, 20, 7), (64, 63, 2), (512, 100, 2), (16, 0, 7), (8, 7, 3)]:
        cf = [felt(SEED + 300 + order, i) for i in range(deg + 1)]
        g = BF2.primitive_nth_root(order)
        vals = refntt.fast_coset_evaluate(Polynomial([B(v) for v in cf]), B(offset), g, order)
        back = refntt.fast_coset_interpolate(B(offset), g, vals)
        ce.append({"order": order, "offset": offset, "coefficients": cf, "values": [v.value for v in vals],
                   "interpolated": [c.value for c in back.coefficients]})
    out["coset"] = ce
    arr = [felt(SEED + 400, i) for i in range(33)]
    out["batch_inverse"] = {"in": arr, "out": [e.value for e in refntt.batch_inverse([B(v) for v in arr])],
                            "zero_message": assertion_message(lambda: refntt.batch_inverse([B(1), B(0)]))}
    # Fri.Domain (fri.py:14-44)
    N = 64
    dom = Fri.Domain(BF.generator(), BF.primitive_nth_root(N), N)
    cf = [felt(SEED + 500, i) for i in range(16)]
    xcf = [[felt(SEED + 600, 3 * i + k) for k in range(3)] for i in range(16)]
    ev = dom.evaluate(Polynomial([B(v, BF) for v in cf]))
    xev = dom.xevaluate(Polynomial([X(l) for l in xcf]))
    out["domain64"] = {"offset": dom.offset.value, "omega": dom.omega.value, "call_5": dom(5).value,
                       "list_head": [e.value for e in dom.list()[:4]],
                       "coefficients": cf, "evaluate": [e.value for e in ev],
                       "interpolate": [c.value for c in dom.interpolate(ev).coefficients],
                       "xcoefficients": xcf, "xevaluate": [xl3(e) for e in xev],
                       "xinterpolate": [xl3(c) for c in dom.xinterpolate(xev).coefficients]}
    dump("poly.json", out)


def gen_poly2():
    """fast_zerofier / fast_evaluate / fast_interpolate (ntt.py:82-161), the rows SURVEY 8f-3 lists as next."""
    out = {"cases": []}
    n = 64
    w = BF2.primitive_nth_root(n)
    for N in (1, 2, 5, 13, 24, 31):
        dom = [felt(SEED + 700 + N, i) for i in range(N)]
        vals = [felt(SEED + 800 + N, i) for i in range(N)]
        D, V = [B(v) for v in dom], [B(v) for v in vals]
        z = refntt.fast_zerofier(D, w, n)
        poly = refntt.fast_interpolate(D, V, w, n)
        ev = refntt.fast_evaluate(poly, D, w, n)
        pc = [felt(SEED + 900 + N, i) for i in range(N + 3)]
        ev2 = refntt.fast_evaluate(Polynomial([B(v) for v in pc]), D, w, n)
        out["cases"].append({"N": N, "root_order": n, "domain": dom, "values": vals, "zerofier": [c.value for c in z.coefficients],
                             "interpolant": [c.value for c in poly.coefficients], "interpolant_degree": poly.degree(),
                             "evaluated_back": [e.value for e in ev], "poly": pc, "poly_evaluated": [e.value for e in ev2]})
    dump("poly2.json", out)


def gen_polyx():
    """The fast_* family over the cubic extension (ntt.py:45-79, 82-161, 177-235 on ExtensionFieldElement operands with a lifted
    root), the way Table.ldex reaches it (table.py:112-149: interpolate_columns lifts omicron powers, one odd power of omega and omega
    itself into the extension field and calls fast_interpolate on extension values)."""
    out = {}
    def XP(seed, n, limbs=3):
        return [[felt(seed, 3 * i + k) if k < limbs else 0 for k in range(3)] for i in range(n)]
    cases = []
    # (root order, deg lhs, deg rhs, limbs lhs, limbs rhs): degree sums 7 (schoolbook, ntt.py:59-60), 8, 63 (no halving at order 64),
    # 64..126 at order 128, halving of a larger order down to the product's size, operands that are lifted base polynomials, zero
    for (n, dl, dr, ll, lr) in [(64, -1, 5, 3, 3), (64, 3, 4, 3, 3), (64, 4, 4, 3, 3), (64, 0, 8, 3, 3), (64, 31, 32, 3, 3), (64, 40, 23, 3, 1),
                                (128, 31, 32, 3, 3), (128, 63, 63, 3, 3), (128, 64, 1, 2, 3), (128, 9, 7, 1, 1), (1024, 20, 13, 3, 3)]:
        w = XF.lift(BF.primitive_nth_root(n))
        lc, rc = XP(SEED + 1100 + dl, dl + 1, ll), XP(SEED + 1200 + dr, dr + 1, lr)
        if (dl, dr) == (40, 23):
            lc = lc + [[0, 0, 0], [0, 0, 0]]          # trailing zero coefficients must not change the product
        lhs, rhs = Polynomial([X(v) for v in lc]), Polynomial([X(v) for v in rc])
        prod = refntt.fast_multiply(lhs, rhs, w, n)
        rec = {"order": n, "lhs": lc, "rhs": rc, "product": [xl3(c) for c in prod.coefficients]}
        if dl >= 0 and dr >= 0:
            q = refntt.fast_coset_divide(prod, lhs, XF.lift(BF.generator()), w, n)
            rec["quotient_by_lhs"] = [xl3(c) for c in q.coefficients]
        cases.append(rec)
    out["fast_multiply"] = cases
    arr = XP(SEED + 1300, 37) + [[5, 0, 0], [0, 9, 0], [0, 0, P - 1], [P - 1, P - 1, P - 1], [1, 0, 0]]
    out["batch_inverse"] = {"in": arr, "out": [xl3(e) for e in refntt.batch_inverse([X(v) for v in arr])],
                            "zero_message": assertion_message(lambda: refntt.batch_inverse([X([1, 2, 3]), X([])]))}
    # the domain shape of table.py:120-124: omicron^i for i < height, then omega^(2 i + 1) for the randomizers
    interp = []
    for (N, height, nrand) in [(64, 8, 1), (128, 16, 1), (256, 32, 1), (64, 4, 2), (128, 1, 1)]:
        omega = BF.primitive_nth_root(N)
        omicron = BF.primitive_nth_root(height) if height > 1 else BF.one()
        dom = [XF.lift(omicron ^ i) for i in range(height)] + [XF.lift(omega ^ (2 * i + 1)) for i in range(nrand)]
        vals = XP(SEED + 1400 + N + height, height + nrand)
        w = XF.lift(omega)
        z = refntt.fast_zerofier(dom, w, N)
        poly = refntt.fast_interpolate(dom, [X(v) for v in vals], w, N)
        back = refntt.fast_evaluate(poly, dom, w, N)
        interp.append({"N": N, "height": height, "num_randomizers": nrand, "domain": [xl3(d) for d in dom], "values": vals,
                       "zerofier": [xl3(c) for c in z.coefficients], "interpolant": [xl3(c) for c in poly.coefficients],
                       "interpolant_degree": poly.degree(), "evaluated_back": [xl3(e) for e in back]})
    out["interpolate_columns"] = interp
    # generic extension-field points (not lifted): the subproduct tree over a domain with all three limbs in use
    N = 64
    w = XF.lift(BF.primitive_nth_root(N))
    dom, vals = XP(SEED + 1500, 19), XP(SEED + 1600, 19)
    poly = refntt.fast_interpolate([X(v) for v in dom], [X(v) for v in vals], w, N)
    pc = XP(SEED + 1700, 25)
    ev = refntt.fast_evaluate(Polynomial([X(v) for v in pc]), [X(v) for v in dom], w, N)
    out["generic_points"] = {"N": N, "domain": dom, "values": vals, "interpolant": [xl3(c) for c in poly.coefficients],
                             "zerofier": [xl3(c) for c in refntt.fast_zerofier([X(v) for v in dom], w, N).coefficients],
                             "poly": pc, "poly_evaluated": [xl3(e) for e in ev]}
    dump("polyx.json", out)


def gen_polyxo():
    """The call surface of ntt.py with arguments that do NOT come from the base field (round-5 verdict, next #7).
    * A transform ROOT must be a 2^k-th root of unity.  p^3 - 1 = (p - 1)(p^2 + p + 1) and p^2 + p + 1 is odd, so the 2-part of the
      cubic extension's multiplicative group is that of the base field: every root of unity of power-of-two order is a lifted base
      element, and an extension element with a non-zero X or X^2 coefficient fails the reference's own assertions -- recorded here.
    * A coset OFFSET may be any non-zero extension element: fast_coset_evaluate / _interpolate / _divide scale by its powers
      (univariate.py:168-169) before / after the transform -- recorded on generic offsets."""
    out = {}
    def XP(seed, n, limbs=3):
        return [[felt(seed, 3 * i + k) if k < limbs else 0 for k in range(3)] for i in range(n)]
    not_a_root = X([felt(SEED + 1801, 0), felt(SEED + 1801, 1), 0])
    vals8 = [X(v) for v in XP(SEED + 1802, 8)]
    w8 = XF.lift(BF.primitive_nth_root(8))
    out["extension_root"] = {
        "root": xl3(not_a_root), "values": [xl3(v) for v in vals8],
        "ntt": assertion_message(lambda: refntt.ntt(not_a_root, vals8)),
        "intt": assertion_message(lambda: refntt.intt(not_a_root, vals8)),
        "fast_multiply": assertion_message(lambda: refntt.fast_multiply(Polynomial(vals8), Polynomial(vals8), not_a_root, 16)),
        "fast_coset_divide": assertion_message(lambda: refntt.fast_coset_divide(Polynomial(vals8), Polynomial(vals8[:3]), w8, not_a_root, 16)),
        "lifted_root_values_ntt": [xl3(v) for v in refntt.ntt(w8, vals8)]}
    cases = []
    for (n, deg, seed) in [(64, 20, 1), (64, 63, 2), (128, 70, 3), (1024, 300, 4)]:
        w = XF.lift(BF.primitive_nth_root(n))
        offset = X(XP(SEED + 1810 + seed, 1)[0])
        pc = XP(SEED + 1820 + seed, deg + 1)
        poly = Polynomial([X(v) for v in pc])
        values = refntt.fast_coset_evaluate(poly, offset, w, n)
        back = refntt.fast_coset_interpolate(offset, w, values)
        rec = {"order": n, "offset": xl3(offset), "poly": pc, "values_sha": sha_xfe_soa(values), "values_head": [xl3(v) for v in values[:4]],
               "interpolated_back": [xl3(c) for c in back.coefficients]}
        if deg < n // 2:
            dc = XP(SEED + 1830 + seed, 7)
            divisor = Polynomial([X(v) for v in dc])
            prod = poly * divisor
            q = refntt.fast_coset_divide(prod, divisor, offset, w, n)
            rec["divisor"], rec["product"], rec["quotient"] = dc, [xl3(c) for c in prod.coefficients], [xl3(c) for c in q.coefficients]
        cases.append(rec)
    out["extension_offset"] = cases
    dump("polyxo.json", out)


def leaf_record(obj, limbs):
    bs = pickle.dumps(obj)
    return {"limbs": limbs, "pickle": bs.hex(), "blake2b": hashlib.blake2b(bs).hexdigest()}


def gen_pickle():
    out = {}
    mags = [0, 1, 7, 255, 256, 300, 65535, 65536, (1 << 31) - 1, 1 << 31, (1 << 32) - 1, 1 << 32, 1 << 39, (1 << 40) - 1,
            1 << 40, (1 << 47) + 5, (1 << 48) - 1, 1 << 55, (1 << 56) - 1, 1 << 56, (1 << 63) - 1, 1 << 63, P - 1,
            felt(SEED, 0), felt(SEED, 1), felt(SEED, 2)]
    out["bfe_leaves"] = [leaf_record(B(v), [v]) for v in mags]
    xs = [[]]
    xs += [[v] for v in mags[1:]]
    xs += [[mags[i], mags[(i * 7 + 3) % len(mags)]] for i in range(len(mags)) if mags[(i * 7 + 3) % len(mags)] != 0]
    xs += [[mags[i], mags[(i * 5 + 1) % len(mags)], mags[(i * 11 + 2) % len(mags)]]
           for i in range(len(mags)) if mags[(i * 11 + 2) % len(mags)] != 0]
    xs += [[felt(SEED + 77, 3 * i + k) for k in range(3)] for i in range(16)]
    xs += [[0, 0, 5], [0, 9, 0], [0, 9], [7, 300, 1 << 40]]
    out["xfe_leaves"] = [leaf_record(X(l), xl(X(l))) for l in xs]
    salt = bytes(range(24))
    out["salt"] = {"salt": salt.hex(), "pickle": pickle.dumps(salt).hex()}
    roots = [hashlib.blake2b(bytes([i])).digest() for i in range(4)]
    out["root_lists"] = [{"roots": [r.hex() for r in roots[:m]], "pickle": pickle.dumps(roots[:m]).hex(),
                          "shake256_32": hashlib.shake_256(pickle.dumps(roots[:m])).digest(32).hex()}
                         for m in range(0, 5)]
    # mixed transcripts: exercises the memo (shared field objects, repeated objects, tuples, nested lists)
    e = [X([felt(SEED + 88, 3 * i + k) for k in range(3)]) for i in range(6)]
    z = X([])
    mixed = []
    objs = [roots[0], roots[1], [e[0], e[1], e[2]]]
    mixed.append(("roots_then_codeword", objs))
    objs = [roots[0], [e[0], e[1], e[2], z], (e[0], e[3], e[1]), [roots[2], roots[3], roots[2]]]
    mixed.append(("shared_objects", objs))
    objs = [(e[4], e[5], e[4]), [roots[1]], (z, X([5]), X([0, 6]))]
    mixed.append(("tuples_first", objs))
    objs = [B(5, BF), roots[0], (B(6, BF), e[0]), [salt, [roots[1]]]]
    mixed.append(("with_bfe_and_salt", objs))
    out["transcripts"] = []
    for name, objs in mixed:
        ps = ProofStream()
        for o in objs:
            ps.push(o)
        out["transcripts"].append({"name": name, "pickle": ps.serialize().hex(),
                                   "fiat_shamir": ps.prover_fiat_shamir().hex()})
    out["transcript_recipes"] = ("roots r_i = blake2b(bytes([i])); e_i = X(felt(SEED+88, 3i..3i+2)); z = X([]); salt = bytes(range(24));"
                                 " roots_then_codeword = [r0, r1, [e0,e1,e2]];"
                                 " shared_objects = [r0, [e0,e1,e2,z], (e0,e3,e1), [r2,r3,r2]];"
                                 " tuples_first = [(e4,e5,e4), [r1], (z, X([5]), X([0,6]))];"
                                 " with_bfe_and_salt = [B(5), r0, (B(6), e0), [salt, [r1]]] with B in the xfield's internal base field")
    # many memo entries: forces LONG_BINGET (index >= 256)
    ps = ProofStream()
    many = [hashlib.blake2b(bytes([i % 256, i // 256, 1])).digest() for i in range(300)]
    for r in many[:150]:
        ps.push(r)
    ps.push([e[0], e[1]])
    for r in many[150:]:
        ps.push(r)
    ps.push((e[0], e[2], e[1]))
    ps.push(many[:3])
    out["long_memo"] = {"pickle_sha256": hashlib.sha256(ps.serialize()).hexdigest(), "length": len(ps.serialize()),
                        "fiat_shamir": ps.prover_fiat_shamir().hex(),
                        "recipe": "m_i = blake2b(bytes([i%256, i//256, 1])) i<300; push m_0..m_149, [e0,e1], m_150..m_299, (e0,e2,e1), [m_0,m_1,m_2]"}
    dump("pickle.json", out)


def gen_merkle():
    out = {}
    trees = []
    for n in (1, 2, 3, 4, 5, 8, 13, 16):
        leaves = [X([felt(SEED + 900 + n, 3 * i + k) for k in range(3)]) for i in range(n)]
        t = Merkle(leaves)
        trees.append({"n": n, "seed_offset": 900 + n, "depth": t.depth, "root": t.root().hex(),
                      "nodes": [x.hex() for x in t.nodes],
                      "paths": [[x.hex() for x in t.open(i)] for i in range(n)]})
    out["xfe_trees"] = trees
    n = 8
    bl = [B(felt(SEED + 950, i)) for i in range(n)]
    t = Merkle(bl)
    out["bfe_tree"] = {"n": n, "seed_offset": 950, "root": t.root().hex(), "nodes": [x.hex() for x in t.nodes]}
    # arbitrary picklable leaves (test_merkle.py:58-63 uses [bytes, bytes] lists)
    gl = [[bytes((splitmix64(i * 1000 + j) & 0xFF) for j in range(splitmix64(i) % 200)),
           bytes((splitmix64(i * 2000 + j) & 0xFF) for j in range(splitmix64(i + 64) % 200))] for i in range(16)]
    t = Merkle(gl)
    out["generic_tree"] = {"n": 16, "root": t.root().hex(),
                           "recipe": "leaf i = [bytes(splitmix64(1000i+j)&255 for j<splitmix64(i)%200), bytes(splitmix64(2000i+j)&255 for j<splitmix64(i+64)%200)]",
                           "path_5": [x.hex() for x in t.open(5)]}
    # salted tree with a deterministic urandom (salted_merkle.py:2,25)
    ctr = [0]

    def fake_urandom(k):
        ctr[0] += 1
        return hashlib.shake_256(b"salt" + ctr[0].to_bytes(8, "little")).digest(k)
    salted_merkle.urandom = fake_urandom
    n = 8
    leaves = [X([felt(SEED + 970, 3 * i + k) for k in range(3)]) for i in range(n)]
    t = SaltedMerkle(leaves)
    s3, p3 = t.open(3)
    out["salted_tree"] = {"n": n, "seed_offset": 970, "salt_recipe": "salt_i = shake256(b'salt' + le64(i+1)).digest(24)",
                          "salts": [l[1].hex() for l in t.leafs], "root": t.root().hex(),
                          "nodes": [x.hex() for x in t.nodes], "open3_salt": s3.hex(), "open3_path": [x.hex() for x in p3],
                          "verify3": SaltedMerkle.verify(t.root(), 3, s3, p3, leaves[3])}
    dump("merkle.json", out)


def run_fri(tag, logd, expansion, t, coeff_fn, keep_stream, prepush=None, disturb=None, write_stream=True):
    d = 1 << logd
    N = d * expansion
    omega = BF.primitive_nth_root(N)
    fri = Fri(BF.generator(), omega, N, expansion, t, XF)
    poly = Polynomial([coeff_fn(i) for i in range(d)])
    t0 = time.perf_counter()
    codeword = fri.domain.xevaluate(poly)
    t1 = time.perf_counter()
    print(tag, "xevaluate %.1fs" % (t1 - t0), flush=True)
    if disturb:
        for i in disturb:
            codeword[i] = XF.zero()
    ps = ProofStream()
    if prepush:
        for o in prepush:
            ps.push(o)
    npre = len(ps.objects)
    root0 = Merkle(codeword).root()
    t2 = time.perf_counter()
    # keep the intermediate codewords prove() computes: wrap commit() (fri.py:183) to capture its result
    cap = {}
    orig_commit = fri.commit

    def capturing_commit(cw, stream, round_index=0):
        res = orig_commit(cw, stream, round_index)
        cap["codewords"], cap["trees"] = res
        return res
    fri.commit = capturing_commit
    idx = fri.prove(codeword, ps)
    t3 = time.perf_counter()
    print(tag, "prove %.1fs" % (t3 - t2), flush=True)
    codewords = cap["codewords"]
    R = fri.num_rounds()
    roots = [root0] + [o for o in ps.objects[npre:npre + R - 1]]
    last = ps.objects[npre + R - 1]
    ser = ps.serialize()
    vs = ProofStream()
    vs.objects = list(ps.objects)
    vs.read_index = npre
    verdict = fri.verify(vs, root0)
    # alphas: recompute as commit() does (fri.py:120) from prefixes of the stream
    alphas = []
    for r in range(R - 1):
        tmp = ProofStream()
        tmp.objects = ps.objects[:npre + r]
        alphas.append(xl3(XF.sample(tmp.prover_fiat_shamir())))
    rec = {"log_degree": logd, "expansion": expansion, "num_colinearity_tests": t, "N": N, "rounds": R,
           "offset": fri.domain.offset.value, "omega": omega.value,
           "codeword_sha": sha_xfe_soa(codeword), "codeword_head": [xl3(e) for e in codeword[:2]],
           "codeword_tail": [xl3(e) for e in codeword[-2:]],
           "codeword_shas": [sha_xfe_soa(c) for c in codewords],
           "roots": [r.hex() for r in roots], "alphas": alphas,
           "last_codeword": [xl(e) for e in last], "indices": idx, "num_objects": len(ps.objects),
           "num_prepushed": npre, "serialize_sha256": hashlib.sha256(ser).hexdigest(), "serialize_len": len(ser),
           "final_fiat_shamir": ps.prover_fiat_shamir().hex(), "verify": bool(verdict),
           "ref_seconds": {"xevaluate": t1 - t0, "prove": t3 - t2}}
    if disturb:
        rec["disturb"] = list(disturb)
    if N <= 1024:
        rec["codeword"] = [xl3(e) for e in codeword]
    if write_stream and keep_stream:
        with open(os.path.join(HERE, "fri_%s_stream.bin" % tag), "wb") as f:
            f.write(ser)
    return rec


def gen_fri():
    out = {}
    sd = lambda i: X([felt(SEED, 3 * i + k) for k in range(3)])
    out["d16_t2"] = run_fri("d16_t2", 4, 4, 2, sd, True)
    out["d64_t8"] = run_fri("d64_t8", 6, 4, 8, sd, True)
    out["d1024_t4"] = run_fri("d1024_t4", 10, 4, 4, sd, True)
    # the reference's own test (test_fri.py:5-59): degree 63, expansion 16, 17 tests, polynomial [xfield(i)]
    out["test_fri_valid"] = run_fri("test_fri_valid", 6, 16, 17, lambda i: XF(i), True)
    out["test_fri_disturbed"] = run_fri("test_fri_disturbed", 6, 16, 17, lambda i: XF(i), True,
                                        disturb=range(0, 63 // 3))
    # non-fresh proof stream (as in brainfuck_stark.py:336 where earlier roots/openings precede FRI)
    r = [hashlib.blake2b(bytes([i])).digest() for i in range(2)]
    e = [X([felt(SEED + 88, 3 * i + k) for k in range(3)]) for i in range(3)]
    out["d16_t2_prepushed"] = run_fri("d16_t2_prepushed", 4, 4, 2, sd, True, prepush=[r[0], (e[0], e[1], e[2]), [r[1]]])
    out["prepush_recipe"] = "prepush = [r0, (e0,e1,e2), [r1]] with r_i = blake2b(bytes([i])), e_i = X(felt(SEED+88, 3i..3i+2))"
    # sample_indices (fri.py:62-86)
    fri = Fri(BF.generator(), BF.primitive_nth_root(64), 64, 4, 2, XF)
    si = []
    for (seed, size, red, num) in [(b"seed", 32, 8, 2), (bytes(32), 1 << 19, 8, 4), (b"\x01" * 32, 512, 64, 17), (b"x", 16, 16, 16)]:
        si.append({"seed": seed.hex(), "size": size, "reduced_size": red, "number": num,
                   "indices": fri.sample_indices(seed, size, red, num)})
    out["sample_indices"] = si
    out["errors"] = {"length_mismatch": assertion_message(lambda: fri.prove([XF.zero()] * 32, ProofStream())),
                     "too_many": assertion_message(lambda: fri.sample_indices(b"s", 32, 8, 9))}
    dump("fri.json", out)


def gen_ntt20():
    logn = 20
    n = 1 << logn
    w = BF2.primitive_nth_root(n)
    vals = [felt(SEED, i) for i in range(n)]
    t0 = time.perf_counter()
    fw = [e.value for e in refntt.ntt(w, [B(v) for v in vals])]
    t1 = time.perf_counter()
    print("ntt 2^20 %.1fs" % (t1 - t0), flush=True)
    iv = [e.value for e in refntt.intt(w, [B(v) for v in vals])]
    t2 = time.perf_counter()
    print("intt 2^20 %.1fs" % (t2 - t1), flush=True)
    dump("ntt20.json", {"logn": logn, "root": w.value, "sha_in": sha_u64(vals), "sha_ntt": sha_u64(fw),
                        "sha_intt": sha_u64(iv), "ntt_head": fw[:4], "ntt_tail": fw[-4:], "intt_head": iv[:4],
                        "intt_tail": iv[-4:], "ref_seconds": [t1 - t0, t2 - t1], "cores": 1})


def gen_fri20():
    sd = lambda i: X([felt(SEED, 3 * i + k) for k in range(3)])
    rec = run_fri("d2p18_t4", 18, 4, 4, sd, False)
    dump("fri20.json", rec)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "small"
    if what == "small":
        gen_field(); gen_ntt(); gen_poly(); gen_poly2(); gen_polyx(); gen_polyxo(); gen_pickle(); gen_merkle(); gen_fri()
    else:
        {"field": gen_field, "ntt": gen_ntt, "poly": gen_poly, "pickle": gen_pickle, "merkle": gen_merkle,
         "fri": gen_fri, "ntt20": gen_ntt20, "fri20": gen_fri20, "poly2": gen_poly2, "polyx": gen_polyx, "polyxo": gen_polyxo}[what]()
