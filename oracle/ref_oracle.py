"""CPU parity oracle for the polynomial hot path of aszepieniec/stark-brainfuck.

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.  The product package (stark_brainfuck_amd) never imports anything from oracle/.

Parity status: PINNED against the reference -- tests/test_oracle_golden.py compares every function
here with tests/golden/*.json, produced by running the reference (tests/golden/gen_golden.py).

Two layers:
  * arithmetic (field, extension field, NTT, coset evaluation, fold) -> oracle/gl_oracle.c via ctypes;
  * commitments / transcript (Merkle, SaltedMerkle, ProofStream, Fri.commit/query/prove) -> restated
    here in Python.  The reference's byte streams are `pickle.dumps(python_object)`; the oracle
    reproduces them with CPython's own `pickle` (protocol 4) applied to *look-alike* objects: classes
    with the reference's module/class/attribute names (algebra.BaseFieldElement, univariate.Polynomial,
    extension_field.ExtensionFieldElement, ...), registered under those module names in sys.modules of
    the test process.  Hashing is CPython's hashlib, exactly as in the reference.  This keeps the oracle
    independent of the product's own pickle emitter / BLAKE2b / Keccak implementations.

All file:line citations are relative to /root/reference/code.
"""
import ctypes
import hashlib
import os
import pickle
import subprocess
import sys
import types

import numpy as np

P = 18446744069414584321          # algebra.py:110-115
GENERATOR = 7                     # algebra.py:117-120
_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgl_oracle.so")
PICKLE_PROTOCOL = 4               # the reference pickles with the interpreter default (3.8-3.13: 4)


def _source_key():
    h = hashlib.sha256()
    for name in ("gl_oracle.c", "Makefile"):
        with open(os.path.join(_HERE, name), "rb") as fh:
            h.update(name.encode() + b"\0" + hashlib.sha256(fh.read()).digest())
    return h.hexdigest()


def build(force=False):
    """compile oracle/gl_oracle.c -> oracle/libgl_oracle.so (gcc, recipe: oracle/Makefile) unless the library was made
    from exactly the present sources -- decided by a content hash kept beside it, not by file times."""
    key = _source_key()
    try:
        current = os.path.exists(_LIB_PATH) and open(_LIB_PATH + ".key").read() == key
    except OSError:
        current = False
    if force or not current:
        subprocess.check_call(["make", "-s", "-B", "-C", _HERE, "libgl_oracle.so"])
        with open(_LIB_PATH + ".key", "w") as fh:
            fh.write(key)
    return _LIB_PATH


def _load():
    build()
    lib = ctypes.CDLL(_LIB_PATH)
    u64, sz, vp = ctypes.c_uint64, ctypes.c_size_t, ctypes.c_void_p
    for name, res, args in [
        ("glo_add", u64, [u64, u64]), ("glo_sub", u64, [u64, u64]), ("glo_mul", u64, [u64, u64]),
        ("glo_neg", u64, [u64]), ("glo_pow", u64, [u64, u64]), ("glo_inv", u64, [u64]),
        ("glo_primitive_nth_root", u64, [ctypes.c_uint32]), ("glo_sample", u64, [ctypes.c_char_p, sz]),
        ("glo_ntt", ctypes.c_int, [u64, vp, vp, sz]), ("glo_intt", ctypes.c_int, [u64, vp, vp, sz]),
        ("glo_scale", None, [u64, vp, vp, sz]),
        ("glo_fast_coset_evaluate", ctypes.c_int, [vp, sz, u64, u64, sz, vp]),
        ("glo_fast_coset_interpolate", ctypes.c_int, [u64, u64, vp, sz, vp]),
        ("glo_batch_inverse", ctypes.c_int, [vp, vp, sz]), ("glo_hadamard", None, [vp, vp, vp, sz]),
        ("xo_add", None, [vp, vp, vp]), ("xo_sub", None, [vp, vp, vp]), ("xo_mul", None, [vp, vp, vp]),
        ("xo_inv", None, [vp, vp]), ("xo_pow", None, [vp, u64, vp]),
        ("xo_fri_fold", None, [vp, sz, vp, u64, u64, vp]),
        ("xo_hadamard", None, [vp, vp, vp, sz]), ("xo_batch_inverse", ctypes.c_int, [vp, vp, sz]),
        ("glo_felt", u64, [u64, u64]), ("glo_felt_fill", None, [u64, u64, vp, sz]),
    ]:
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


_lib = _load()

_NTT_ERRORS = {  # messages of the reference's asserts (ntt.py:5-6, 13-16, 28-35)
    1: "cannot compute ntt of non-power-of-two sequence",
    2: "primitive root must be nth root of unity",
    3: "is not primitive nth root of unity",
    4: "more coefficients than the evaluation order",
    5: "batch inverse does not work when input contains a zero",
}


def _arr(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.uint64))


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _x3(limbs):
    l = [int(v) % P for v in limbs]
    return (ctypes.c_uint64 * 3)(*(l + [0] * (3 - len(l))))


# ------------------------------------------------------------------ inputs
def felt(seed, i):
    return int(_lib.glo_felt(seed & 0xFFFFFFFFFFFFFFFF, i))


def felt_array(seed, start, n):
    out = np.empty(n, dtype=np.uint64)
    _lib.glo_felt_fill(seed & 0xFFFFFFFFFFFFFFFF, start, _ptr(out), n)
    return out


# ------------------------------------------------------------------ base field (algebra.py)
def add(a, b): return int(_lib.glo_add(a, b))
def sub(a, b): return int(_lib.glo_sub(a, b))
def mul(a, b): return int(_lib.glo_mul(a, b))
def neg(a): return int(_lib.glo_neg(a))
def power(a, e): return int(_lib.glo_pow(a, e))
def inv(a): return int(_lib.glo_inv(a))
def primitive_nth_root(n): return int(_lib.glo_primitive_nth_root(int(n).bit_length() - 1))
def sample(bs): return int(_lib.glo_sample(bytes(bs), len(bs)))


# ------------------------------------------------------------------ extension field (extension_field.py)
def xadd(a, b):
    r = (ctypes.c_uint64 * 3)(); _lib.xo_add(_x3(a), _x3(b), r); return list(r)


def xsub(a, b):
    r = (ctypes.c_uint64 * 3)(); _lib.xo_sub(_x3(a), _x3(b), r); return list(r)


def xmul(a, b):
    r = (ctypes.c_uint64 * 3)(); _lib.xo_mul(_x3(a), _x3(b), r); return list(r)


def xinv(a):
    r = (ctypes.c_uint64 * 3)(); _lib.xo_inv(_x3(a), r); return list(r)


def xpow(a, e):
    r = (ctypes.c_uint64 * 3)(); _lib.xo_pow(_x3(a), e, r); return list(r)


def xsample(bs):
    """extension_field.py:100-111: three chunks of len//3 bytes, each sampled big-endian mod p."""
    bs = bytes(bs)
    c = len(bs) // 3
    return [sample(bs[i * c:(i + 1) * c]) for i in range(3)]


def xtrim(limbs):
    """stored form of an extension element: trailing zero coefficients dropped (extension_field.py:6-9)."""
    l = [int(v) for v in limbs]
    while l and l[-1] == 0:
        l.pop()
    return l


# ------------------------------------------------------------------ NTT family (ntt.py), array in / array out
def _check(rc):
    if rc:
        raise AssertionError(_NTT_ERRORS.get(rc, "oracle error %d" % rc))


def ntt(root, values):
    v = _arr(values); out = np.empty_like(v)
    _check(_lib.glo_ntt(root, _ptr(v), _ptr(out), v.size)); return out


def intt(root, values):
    v = _arr(values); out = np.empty_like(v)
    _check(_lib.glo_intt(root, _ptr(v), _ptr(out), v.size)); return out


# ---- the reference's own cost model, for bench.py's cpu_baseline: ntt.py:4-23 on BOXED elements in pure Python ----
# The reference cannot travel to the GPU box, so its CPU figure there is taken from this restatement: every field element is a
# Python object holding an int and its field, every + and * goes through a method of the field object and allocates a new
# element (algebra.py:12-19, 89-99), a power is square-and-multiply over the bits of the exponent (algebra.py:34-41), and the
# transform is the recursive radix-2 of ntt.py:4-23 including its two asserts and the per-index power `primitive_root ^ i` --
# the same operation count, allocation pattern and interpreter dispatch as the reference, on the same CPython.
class _BoxedField:
    def __init__(self, p):
        self.p = p

    def sum(self, a, b):
        return _Boxed((a.value + b.value) % self.p, self)

    def product(self, a, b):
        return _Boxed((a.value * b.value) % self.p, self)


class _Boxed:
    def __init__(self, value, field):
        self.value = value
        self.field = field

    def __add__(self, other):
        return self.field.sum(self, other)

    def __mul__(self, other):
        return self.field.product(self, other)

    def __xor__(self, e):                        # algebra.py:34-41
        result, base = _Boxed(1, self.field), _Boxed(self.value, self.field)
        for bit in reversed(range(len(bin(e)[2:]))):
            result = result * result
            if (e >> bit) & 1:
                result = result * base
        return result


def _ntt_boxed(w, xs):
    n = len(xs)
    if n <= 1:
        return xs
    assert (w ^ n).value == 1, f"primitive root must be nth root of unity, where n is {n}"
    assert (w ^ (n // 2)).value != 1, "primitive root is not primitive nth root of unity"
    h = n // 2
    w2 = w ^ 2
    lo, hi = _ntt_boxed(w2, xs[::2]), _ntt_boxed(w2, xs[1::2])       # the reference transforms the odd half first; same values
    return [lo[k % h] + (w ^ k) * hi[k % h] for k in range(n)]


def ntt_python(root, values):
    """ntt.py:4-23 in pure Python on boxed elements (see above); ints in, ints out"""
    assert len(values) & (len(values) - 1) == 0, "cannot compute ntt of non-power-of-two sequence"
    f = _BoxedField(P)
    return [e.value for e in _ntt_boxed(_Boxed(int(root), f), [_Boxed(int(v), f) for v in values])]


def scale(factor, coeffs):
    v = _arr(coeffs); out = np.empty_like(v)
    _lib.glo_scale(factor, _ptr(v), _ptr(out), v.size); return out


def fast_coset_evaluate(coeffs, offset, generator, order):
    v = _arr(coeffs); out = np.empty(order, dtype=np.uint64)
    _check(_lib.glo_fast_coset_evaluate(_ptr(v), v.size, offset, generator, order, _ptr(out))); return out


def fast_coset_interpolate(offset, generator, values):
    v = _arr(values); out = np.empty_like(v)
    _check(_lib.glo_fast_coset_interpolate(offset, generator, _ptr(v), v.size, _ptr(out))); return out


def batch_inverse(values):
    v = _arr(values); out = np.empty_like(v)
    _check(_lib.glo_batch_inverse(_ptr(v), _ptr(out), v.size)); return out


def hadamard(a, b):
    a, b = _arr(a), _arr(b); out = np.empty_like(a)
    _lib.glo_hadamard(_ptr(a), _ptr(b), _ptr(out), a.size); return out


def _degree(c):
    d = len(c) - 1
    while d >= 0 and int(c[d]) == 0:
        d -= 1
    return d


def _schoolbook(l, r):
    """univariate.py:40-51"""
    if len(l) == 0 or len(r) == 0:
        return []
    out = [0] * (len(l) + len(r) - 1)
    for i, a in enumerate(l):
        for j, b in enumerate(r):
            out[i + j] = add(out[i + j], mul(int(a), int(b)))
    return out


def fast_multiply(lhs, rhs, root, order):
    """ntt.py:45-79, coefficient lists in / out."""
    assert power(root, order) == 1, "supplied root does not have supplied order"
    assert power(root, order // 2) != 1, "supplied root is not primitive root of supplied order"
    dl, dr = _degree(lhs), _degree(rhs)
    if dl < 0 or dr < 0:
        return []
    degree = dl + dr
    if degree < 8:
        return _schoolbook(list(lhs), list(rhs))          # ntt.py:59-60 (no trimming of lhs*rhs)
    while degree < order // 2:
        root, order = mul(root, root), order // 2         # ntt.py:62-64
    a = np.zeros(order, dtype=np.uint64); a[:dl + 1] = _arr(lhs)[:dl + 1]
    b = np.zeros(order, dtype=np.uint64); b[:dr + 1] = _arr(rhs)[:dr + 1]
    prod = intt(root, hadamard(ntt(root, a), ntt(root, b)))
    return [int(v) for v in prod[:degree + 1]]


def xntt_soa(root, limbs_soa):
    """NTT over the extension field with a lifted base root = three limb NTTs (ntt.py:11, fri.py:37)."""
    return np.stack([ntt(root, limbs_soa[k]) for k in range(3)])


def xintt_soa(root, limbs_soa):
    return np.stack([intt(root, limbs_soa[k]) for k in range(3)])


def _xsoa(elems, n=None):
    """list of 3-limb elements -> (3, n) limb-major array, zero padded to n"""
    n = len(elems) if n is None else n
    soa = np.zeros((3, n), dtype=np.uint64)
    for i, e in enumerate(elems):
        for k, v in enumerate(e):
            soa[k, i] = int(v)
    return soa


def _xlist(soa, count=None):
    count = soa.shape[1] if count is None else count
    return [[int(soa[0, i]), int(soa[1, i]), int(soa[2, i])] for i in range(count)]


def xhadamard(a_soa, b_soa):
    a, b = np.ascontiguousarray(a_soa, dtype=np.uint64), np.ascontiguousarray(b_soa, dtype=np.uint64)
    out = np.empty_like(a)
    _lib.xo_hadamard(_ptr(a), _ptr(b), _ptr(out), a.shape[1]); return out


def xbatch_inverse(soa):
    """ntt.py:177-188 on extension elements, (3, n) in / out"""
    a = np.ascontiguousarray(soa, dtype=np.uint64); out = np.empty_like(a)
    _check(_lib.xo_batch_inverse(_ptr(a), _ptr(out), a.shape[1])); return out


def _xdegree(c):
    d = len(c) - 1
    while d >= 0 and not any(int(v) for v in c[d]):
        d -= 1
    return d


def _xschoolbook(l, r):
    """univariate.py:40-51 over the extension field"""
    if len(l) == 0 or len(r) == 0:
        return []
    out = [[0, 0, 0] for _ in range(len(l) + len(r) - 1)]
    for i, a in enumerate(l):
        for j, b in enumerate(r):
            out[i + j] = xadd(out[i + j], xmul(a, b))
    return out


def xfast_multiply(lhs, rhs, root, order):
    """ntt.py:45-79 on polynomials with ExtensionFieldElement coefficients (lists of 3 limbs) and a lifted root of unity
    (table.py:133-134): three limb transforms per operand, the Hadamard product in the extension field, three inverse ones."""
    assert power(root, order) == 1, "supplied root does not have supplied order"
    assert power(root, order // 2) != 1, "supplied root is not primitive root of supplied order"
    dl, dr = _xdegree(lhs), _xdegree(rhs)
    if dl < 0 or dr < 0:
        return []
    degree = dl + dr
    if degree < 8:
        return _xschoolbook(list(lhs), list(rhs))          # ntt.py:59-60: lhs * rhs with the operands as given
    while degree < order // 2:
        root, order = mul(root, root), order // 2
    a, b = _xsoa(lhs[:dl + 1], order), _xsoa(rhs[:dr + 1], order)
    prod = xintt_soa(root, xhadamard(xntt_soa(root, a), xntt_soa(root, b)))
    return _xlist(prod, degree + 1)


def xfast_coset_divide(lhs, rhs, offset, root, order):
    """ntt.py:191-235 on extension polynomials whose division is exact, degree >= 8 (below that the reference does a long division);
    offset and root are base-field values (lifted in the reference)."""
    assert power(root, order) == 1, "supplied root does not have supplied order"
    assert power(root, order // 2) != 1, "supplied root is not primitive root of supplied order"
    dl, dr = _xdegree(lhs), _xdegree(rhs)
    assert dr >= 0, "cannot divide by zero polynomial"
    if dl < 0:
        return []
    assert dr <= dl, "cannot divide by polynomial of larger degree"
    degree = max(dl, dr)
    assert degree >= 8, "oracle restates the transform branch only"
    while degree < order // 2:
        root, order = mul(root, root), order // 2
    a = np.stack([fast_coset_evaluate(_xsoa(lhs[:dl + 1])[k], offset, root, order) for k in range(3)])
    b = np.stack([fast_coset_evaluate(_xsoa(rhs[:dr + 1])[k], offset, root, order) for k in range(3)])
    quo = xintt_soa(root, xhadamard(a, xbatch_inverse(b)))
    quo = np.stack([scale(inv(offset), quo[k]) for k in range(3)])
    return _xlist(quo, dl - dr + 1)


def xevaluate_soa(coeff_soa, offset, omega, length):
    """Fri.Domain.xevaluate (fri.py:32-37) on an SoA coefficient array of shape (3, d)."""
    return np.stack([fast_coset_evaluate(coeff_soa[k], offset, omega, length) for k in range(3)])


def fri_fold(cw_soa, alpha, offset, omega):
    """fri.py:127-128 on an SoA (3, n) codeword -> (3, n/2)."""
    cw = np.ascontiguousarray(cw_soa, dtype=np.uint64)
    n = cw.shape[1]
    out = np.empty((3, n // 2), dtype=np.uint64)
    _lib.xo_fri_fold(_ptr(cw), n, _x3(alpha), offset, omega, _ptr(out))
    return out


# ------------------------------------------------------------------ look-alike objects for pickle parity
def _lookalike_modules():
    """Create classes named exactly like the reference's (module, qualname, attribute order) so that
    CPython's pickle emits the reference's byte stream for them.  Registered in sys.modules because
    pickle verifies that `module.qualname` resolves to the class being pickled."""
    if "algebra" in sys.modules and getattr(sys.modules["algebra"], "_bfs_oracle_lookalike", False):
        m = sys.modules
        return m["algebra"], m["univariate"], m["extension_field"]
    for name in ("algebra", "univariate", "extension_field"):
        if name in sys.modules:
            raise RuntimeError("module %r already imported; the oracle needs that name for its look-alike classes" % name)
    alg, uni, ext = types.ModuleType("algebra"), types.ModuleType("univariate"), types.ModuleType("extension_field")

    class BaseField:                       # algebra.py:76-78 (state: p)
        def __init__(self, p): self.p = p

    class BaseFieldElement:                # algebra.py:15-18 (state: value, field)
        def __init__(self, value, field): self.value = value; self.field = field

    class Polynomial:                      # univariate.py:4-6 (state: coefficients)
        def __init__(self, coefficients): self.coefficients = list(coefficients)

    class ExtensionField:                  # extension_field.py:55-57 (state: modulus)
        def __init__(self, modulus): self.modulus = modulus

    class ExtensionFieldElement:           # extension_field.py:5-9 (state: polynomial, field)
        def __init__(self, polynomial, field): self.polynomial = polynomial; self.field = field

    for mod, classes in ((alg, (BaseField, BaseFieldElement)), (uni, (Polynomial,)), (ext, (ExtensionField, ExtensionFieldElement))):
        for c in classes:
            c.__module__ = mod.__name__
            c.__qualname__ = c.__name__
            setattr(mod, c.__name__, c)
        mod._bfs_oracle_lookalike = True
        sys.modules[mod.__name__] = mod
    return alg, uni, ext


_alg, _uni, _ext = _lookalike_modules()
# ExtensionField.main() (extension_field.py:88-98): modulus [1, p-1, 0, 1] with the SAME `one` object at
# index 0 and 3, every coefficient pointing at one BaseField instance.
_BF = _alg.BaseField(P)
_one = _alg.BaseFieldElement(1, _BF)
_XF = _ext.ExtensionField(_uni.Polynomial([_one, _alg.BaseFieldElement(P - 1, _BF), _alg.BaseFieldElement(0, _BF), _one]))
_BF_STANDALONE = _alg.BaseField(P)     # BaseField.main() called on its own (algebra.py:110-115)


def make_xfe(limbs):
    """variant-A extension element object (coefficients reference the xfield's internal BaseField)."""
    return _ext.ExtensionFieldElement(_uni.Polynomial([_alg.BaseFieldElement(int(v), _BF) for v in xtrim(limbs)]), _XF)


def make_bfe(value, internal=False):
    return _alg.BaseFieldElement(int(value), _BF if internal else _BF_STANDALONE)


def xfe_limbs(obj):
    c = [b.value for b in obj.polynomial.coefficients]
    return c + [0] * (3 - len(c))


def dumps(obj):
    return pickle.dumps(obj, protocol=PICKLE_PROTOCOL)


# ------------------------------------------------------------------ Merkle (merkle.py), SaltedMerkle (salted_merkle.py)
class MerkleOracle:
    """merkle.py:8-52.  `leaf_bytes` are the leaf preimages (already pickled)."""

    def __init__(self, leaf_bytes):
        n = len(leaf_bytes)
        self.num_leafs = n
        npo2 = 1
        while npo2 < n:
            npo2 <<= 1
        if n == 0:
            npo2 = 0
        # merkle.py:9-20: for n = 0 the reference computes next_power_of_two = 0 and depth = 0
        self.depth = max(npo2.bit_length() - 1, 0)
        self.nodes = [bytes(32)] * (2 * npo2)             # merkle.py:26 (32 zero bytes, sic)
        for i, b in enumerate(leaf_bytes):
            self.nodes[npo2 + i] = hashlib.blake2b(b).digest()   # merkle.py:29-32
        for i in range(npo2 - 1, -1, -1):                 # merkle.py:35-41 (also fills the junk node 0)
            self.nodes[i] = hashlib.blake2b(self.nodes[2 * i] + self.nodes[2 * i + 1]).digest()

    def root(self):
        return self.nodes[1]

    def open(self, index):
        path = []
        idx = (1 << self.depth) | index
        while idx > 1:
            path.append(self.nodes[idx ^ 1])
            idx >>= 1
        return path


def merkle_verify(root, index, path, leaf_bytes):
    """merkle.py:54-63"""
    h = hashlib.blake2b(leaf_bytes).digest()
    for node in path:
        h = hashlib.blake2b(h + node).digest() if index % 2 == 0 else hashlib.blake2b(node + h).digest()
        index >>= 1
    return h == root


def xfe_merkle(cw_soa):
    """Merkle tree over an SoA (3, n) extension codeword; leaves pickled one by one like merkle.py:30."""
    n = cw_soa.shape[1]
    objs = [make_xfe([cw_soa[0, i], cw_soa[1, i], cw_soa[2, i]]) for i in range(n)]
    return MerkleOracle([dumps(o) for o in objs]), objs


def salted_leaf_bytes(element_obj, salt):
    """salted_merkle.py:32-35: two separate pickles concatenated."""
    return dumps(element_obj) + dumps(salt)


# ------------------------------------------------------------------ ProofStream (ip.py)
class ProofStreamOracle:
    def __init__(self):
        self.objects = []
        self.read_index = 0

    def push(self, obj):
        self.objects.append(obj)

    def serialize(self):                                  # ip.py:18-19
        return dumps(self.objects)

    def prover_fiat_shamir(self, num_bytes=32):           # ip.py:21-22
        return hashlib.shake_256(self.serialize()).digest(num_bytes)


# ------------------------------------------------------------------ FRI prover (fri.py:54-199)
def fri_num_rounds(length, expansion_factor):             # fri.py:54-60
    r = 0
    while length > expansion_factor:
        length //= 2
        r += 1
    return r


def sample_indices(seed, size, reduced_size, number):     # fri.py:62-86
    assert number <= reduced_size, \
        f"cannot sample more indices than available in last codeword; requested: {number}, available: {reduced_size}"
    indices, reduced = [], []
    counter = 0
    while len(indices) < number:
        digest = hashlib.blake2b(seed + bytes(counter)).digest()    # bytes(counter) = `counter` zero bytes
        index = int.from_bytes(digest, "big") % size
        counter += 1
        if index % reduced_size not in reduced:
            indices.append(index)
            reduced.append(index % reduced_size)
    return indices


def fri_prove(cw_soa, offset, omega, expansion_factor, num_colinearity_tests, proof_stream=None):
    """Fri.prove (fri.py:178-199) with commit (91-139), query (141-158), query_last (160-176).
    cw_soa: (3, N) uint64.  Returns a dict with everything the parity tests compare."""
    ps = proof_stream if proof_stream is not None else ProofStreamOracle()
    cw = np.ascontiguousarray(cw_soa, dtype=np.uint64)
    N = cw.shape[1]
    R = fri_num_rounds(N, expansion_factor)
    assert R >= 1, "cannot do FRI with less than one round"
    t = num_colinearity_tests
    codewords, trees, leaf_objs, roots, alphas = [], [], [], [], []
    w, g = omega, offset
    for r in range(R):                                    # fri.py:100
        n = cw.shape[1]
        assert power(w, n - 1) == inv(w), "error in commit: omega does not have the right order!"
        tree, objs = xfe_merkle(cw)                       # fri.py:108
        roots.append(tree.root())
        if r > 0:
            ps.push(tree.root())                          # fri.py:112-113
        if r == R - 1:
            break
        alpha = xsample(ps.prover_fiat_shamir())          # fri.py:120
        alphas.append(alpha)
        codewords.append(cw); trees.append(tree); leaf_objs.append(objs)
        cw = fri_fold(cw, alpha, g, w)                    # fri.py:127-128
        w, g = mul(w, w), mul(g, g)                       # fri.py:130-131
    last_objs = objs                                      # the last codeword's element objects
    ps.push(last_objs)                                    # fri.py:134  (same objects re-used by query_last)
    codewords.append(cw)
    top = sample_indices(ps.prover_fiat_shamir(), codewords[1].shape[1], codewords[-1].shape[1], t)  # fri.py:186-187
    indices = list(top)
    for i in range(len(trees) - 1):                       # fri.py:191-194
        half = codewords[i].shape[1] // 2
        indices = [x % half for x in indices]
        a_idx, b_idx = indices, [x + half for x in indices]
        for s in range(t):
            ps.push((leaf_objs[i][a_idx[s]], leaf_objs[i][b_idx[s]], leaf_objs[i + 1][indices[s]]))
        for s in range(t):
            ps.push(trees[i].open(a_idx[s])); ps.push(trees[i].open(b_idx[s])); ps.push(trees[i + 1].open(indices[s]))
    last_len = codewords[-1].shape[1]
    indices = [x % last_len for x in indices]             # fri.py:195-197
    half = len(leaf_objs[-1]) // 2
    a_idx, b_idx = indices, [x + half for x in indices]
    for s in range(t):
        ps.push((leaf_objs[-1][a_idx[s]], leaf_objs[-1][b_idx[s]], last_objs[indices[s]]))
    for s in range(t):
        ps.push(trees[-1].open(a_idx[s])); ps.push(trees[-1].open(b_idx[s]))
    return {"roots": roots, "alphas": alphas, "codewords": codewords, "last_codeword": codewords[-1],
            "indices": top, "proof_stream": ps, "rounds": R}
