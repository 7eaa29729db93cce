"""Fast polynomial arithmetic on the GPU -- the mirror of the reference's `ntt.py` (/root/reference/code/ntt.py).

Same functions, argument order, results and AssertionErrors:

    ntt(primitive_root, values)            ntt.py:4-23      intt(primitive_root, values)            ntt.py:26-42
    fast_multiply(lhs, rhs, root, order)   ntt.py:45-79     fast_coset_evaluate(poly, offset, g, n) ntt.py:164-168
    fast_coset_interpolate(offset, g, v)   ntt.py:171-174   batch_inverse(array)                    ntt.py:177-188
    fast_coset_divide(l, r, offset, g, n)  ntt.py:191-235

`values` may be a Python list of BaseFieldElement / ExtensionFieldElement objects (a new list of new objects comes
back, as in the reference) or a BaseArray / XArray living in HBM (an array of the same kind comes back, nothing is
copied to the host).  Every transform runs in the HIP kernels behind bfs_gl_ntt; there is no CPU fallback.
"""
import numpy as np

from . import _lib
from .algebra import BaseFieldElement
from .arrays import BaseArray, XArray, raw_ntt
from .device import current_stream
from .extension_field import ExtensionFieldElement
from .univariate import Polynomial


def _is_genuine_extension(x):
    """an ExtensionFieldElement with a non-zero X or X^2 coefficient"""
    return isinstance(x, ExtensionFieldElement) and any(c.value % _P for c in x.polynomial.coefficients[1:])


def _base_value(x, not_a_root=None):
    """int value of a root / offset given as int, BaseFieldElement or a lifted ExtensionFieldElement.

    A transform ROOT that is a genuine extension element cannot be what the reference's assertions ask for: p^3 - 1 =
    (p - 1)(p^2 + p + 1) with p^2 + p + 1 odd, so every element of power-of-two order of the cubic extension lies in the base field
    and `root ^ n == one` fails for anything else (ntt.py:11-12, 29-30, 46-47, 192-193).  `not_a_root` is that assertion's message
    (tests/golden/polyxo.json has the reference's).  Coset OFFSETS may be any extension element: _offset below."""
    if isinstance(x, int):
        return x
    if isinstance(x, BaseFieldElement):
        return x.value
    if isinstance(x, ExtensionFieldElement):
        if _is_genuine_extension(x):
            raise AssertionError(not_a_root or "supplied root does not have supplied order")
        c = x.polynomial.coefficients
        return c[0].value if c else 0
    raise TypeError("not a field element: %r" % type(x))


_P = (1 << 64) - (1 << 32) + 1


def _offset_powers(offset, count, inverse=False):
    """XArray of offset^k (offset^-k), k < count, for a genuine extension offset: Polynomial.scale (univariate.py:168-169) multiplies
    coefficient k by factor^k.  The powers are a short host loop over this package's own field arithmetic (a correct, slow path: the
    reference's callers only ever pass lifted base offsets, which ride in the transform's first pass instead)."""
    step = offset.inverse() if inverse else offset
    acc = offset.field.one()
    soa = np.zeros((3, count), dtype=np.uint64)
    for k in range(count):
        limbs = acc.limbs()
        soa[0, k], soa[1, k], soa[2, k] = limbs[0], limbs[1], limbs[2]
        acc = acc * step
    return XArray.from_numpy(soa, offset.field)


def _scale_by_powers(arr, offset, count, inverse=False):
    """arr[k] *= offset^(+-k) in place, k < count (arr: XArray)"""
    pw = _offset_powers(offset, count, inverse)
    _lib.check(_lib.load().bfs_xfe_mul_pointwise(arr.ptr, arr.stride, pw.ptr, pw.stride, arr.ptr, arr.stride, count, current_stream()))


def _log2(n):
    assert n & (n - 1) == 0, "cannot compute ntt of non-power-of-two sequence"
    return n.bit_length() - 1


def _is_x(values):
    return isinstance(values, XArray) or (isinstance(values, list) and len(values) and isinstance(values[0], ExtensionFieldElement))


def _to_array(values):
    if isinstance(values, (BaseArray, XArray)):
        return values
    return XArray.from_elements(values) if _is_x(values) else BaseArray.from_elements(values)


def _transform(arr, n_in, n, root, shift, post_scale):
    """run bfs_gl_ntt on a BaseArray (batch rows) or XArray (3 limb planes) -> new array of length n."""
    log_n = _log2(n)
    if isinstance(arr, XArray):
        out = XArray.empty(n, arr.field)
        raw_ntt(arr.ptr, n_in, arr.stride, out.ptr, n, log_n, 3, root, shift, post_scale)
    else:
        out = BaseArray.empty(n, arr.field, arr.batch)
        raw_ntt(arr.ptr, n_in, arr.n, out.ptr, n, log_n, arr.batch, root, shift, post_scale)
    return out


def _pow(v, e):
    return _lib.load().bfs_gl_pow(v, e)


def ntt(primitive_root, values):
    n = len(values)
    assert n & (n - 1) == 0, "cannot compute ntt of non-power-of-two sequence"
    if n <= 1:
        return values                                       # ntt.py:8-9 returns its argument
    as_list = isinstance(values, list)
    out = _transform(_to_array(values), n, n, _base_value(primitive_root, "primitive root must be nth root of unity, where n is %d" % n), 1, 1)
    return out.to_elements() if as_list else out


def intt(primitive_root, values):
    n = len(values)
    assert n & (n - 1) == 0, "cannot compute intt of non-power-of-two sequence"
    w = _base_value(primitive_root)
    assert _pow(w, n) == 1, "supplied root does not have supplied order"
    if n == 1:
        return values
    assert _pow(w, n // 2) != 1, "supplied root is not primitive root of supplied order"
    lib = _lib.load()
    as_list = isinstance(values, list)
    out = _transform(_to_array(values), n, n, lib.bfs_gl_inv(w), 1, lib.bfs_gl_inv(n))
    return out.to_elements() if as_list else out


def _check_root(primitive_root, root_order):
    w = _base_value(primitive_root)
    assert _pow(w, root_order) == 1, "supplied root does not have supplied order"
    assert _pow(w, root_order // 2) != 1, "supplied root is not primitive root of supplied order"
    return w


def _is_xlist(coeffs):
    return bool(coeffs) and isinstance(coeffs[0], ExtensionFieldElement)


def _coeff_array(coeffs, as_x=None):
    """device array of a coefficient list: an XArray when the coefficients are ExtensionFieldElements or `as_x` asks for the
    lifted form (a base polynomial next to an extension one: extension_field.py:113-116), else a BaseArray."""
    if as_x is None:
        as_x = _is_xlist(coeffs)
    if not as_x:
        return BaseArray.from_elements(coeffs)
    if _is_xlist(coeffs):
        return XArray.from_elements(coeffs)
    soa = np.zeros((3, len(coeffs)), dtype=np.uint64)
    soa[0] = np.fromiter((e.value for e in coeffs), dtype=np.uint64, count=len(coeffs))
    return XArray.from_numpy(soa)


def _hadamard(lc, rc, order):
    """lc <- lc * rc point by point (ntt.py:76), base field or cubic extension."""
    lib = _lib.load()
    if isinstance(lc, XArray):
        _lib.check(lib.bfs_xfe_mul_pointwise(lc.ptr, lc.stride, rc.ptr, rc.stride, lc.ptr, lc.stride, order, current_stream()))
    else:
        _lib.check(lib.bfs_gl_mul_pointwise(lc.ptr, rc.ptr, lc.ptr, order, current_stream()))


def _inverse_in_place(arr, count):
    lib = _lib.load()
    if isinstance(arr, XArray):
        _lib.check(lib.bfs_xfe_batch_inverse(arr.ptr, arr.stride, arr.ptr, arr.stride, count, current_stream()))
    else:
        _lib.check(lib.bfs_gl_batch_inverse(arr.ptr, arr.ptr, count, current_stream()))


def fast_multiply(lhs, rhs, primitive_root, root_order):
    w = _check_root(primitive_root, root_order)
    if lhs.is_zero() or rhs.is_zero():
        return Polynomial([])
    degree = lhs.degree() + rhs.degree()
    if degree < 8:
        return lhs * rhs                                    # ntt.py:59-60
    lib = _lib.load()
    order = root_order
    while degree < order // 2:
        w, order = lib.bfs_gl_mul(w, w), order // 2
    lco, rco = lhs.coefficients[:lhs.degree() + 1], rhs.coefficients[:rhs.degree() + 1]
    as_x = _is_xlist(lco) or _is_xlist(rco)                 # extension operands: three limb planes per transform (table.py:133-134)
    la, ra = _coeff_array(lco, as_x), _coeff_array(rco, as_x)
    lc = _transform(la, la.n, order, w, 1, 1)
    rc = _transform(ra, ra.n, order, w, 1, 1)
    _hadamard(lc, rc, order)
    prod = _transform(lc, order, order, lib.bfs_gl_inv(w), 1, lib.bfs_gl_inv(order))
    return Polynomial(prod.to_elements()[:degree + 1])


def fast_coset_evaluate(polynomial, offset, generator, order):
    coeffs = polynomial.coefficients
    assert len(coeffs) <= order, "polynomial has more coefficients than the evaluation domain has points"
    if not coeffs:
        return [offset.field.zero() for _ in range(order)]  # ntt of `order` zeros (ntt.py:166-167)
    if _is_genuine_extension(offset):
        src = _coeff_array(coeffs, True)
        _scale_by_powers(src, offset, len(coeffs))
        return _transform(src, len(coeffs), order, _base_value(generator, "primitive root must be nth root of unity, where n is %d" % order), 1, 1).to_elements()
    src = _coeff_array(coeffs)
    out = _transform(src, len(coeffs), order, _base_value(generator, "primitive root must be nth root of unity, where n is %d" % order), _base_value(offset), 1)
    return out.to_elements()


def fast_coset_interpolate(offset, generator, values):
    n = len(values)
    assert n & (n - 1) == 0, "cannot compute intt of non-power-of-two sequence"
    w = _base_value(generator)
    assert _pow(w, n) == 1, "supplied root does not have supplied order"
    as_list = isinstance(values, list)
    if n == 1:
        coeffs = values if as_list else values.to_elements()
        return Polynomial(coeffs)
    assert _pow(w, n // 2) != 1, "supplied root is not primitive root of supplied order"
    lib = _lib.load()
    # intt followed by scale(offset^-1)  ==  one inverse transform whose outputs are multiplied by offset^-k:
    # done as intt, then a coset "scale" pass (bfs_gl_scale)
    arr = _to_array(values)
    out = _transform(arr, n, n, lib.bfs_gl_inv(w), 1, lib.bfs_gl_inv(n))
    if _is_genuine_extension(offset):
        if not isinstance(out, XArray):
            out = _coeff_array(out.to_elements(), True)
        _scale_by_powers(out, offset, n, inverse=True)
        return Polynomial(out.to_elements())
    oinv = lib.bfs_gl_inv(_base_value(offset))
    batch = 3 if isinstance(out, XArray) else out.batch
    _lib.check(lib.bfs_gl_scale(out.ptr, out.ptr, n, n, batch, oinv, current_stream()))
    return Polynomial(out.to_elements())


def batch_inverse(array):
    if isinstance(array, BaseArray):
        out = BaseArray.empty(array.n, array.field, array.batch)
        _lib.check(_lib.load().bfs_gl_batch_inverse(array.ptr, out.ptr, array.n * array.batch, current_stream()))
        return out
    if isinstance(array, XArray):
        out = XArray.empty(array.n, array.field)
        _lib.check(_lib.load().bfs_xfe_batch_inverse(array.ptr, array.stride, out.ptr, out.stride, array.n, current_stream()))
        return out
    assert all(not a.is_zero() for a in array), "batch inverse does not work when input contains a zero"
    if not array:
        return []
    src = _coeff_array(array)
    _inverse_in_place(src, src.n)
    return src.to_elements()


def fast_coset_divide(lhs, rhs, offset, primitive_root, root_order):
    """exact division on a coset (ntt.py:191-235)."""
    w = _check_root(primitive_root, root_order)
    assert not rhs.is_zero(), "cannot divide by zero polynomial"
    if lhs.is_zero():
        return Polynomial([])
    assert rhs.degree() <= lhs.degree(), "cannot divide by polynomial of larger degree"
    degree = max(lhs.degree(), rhs.degree())
    if degree < 8:
        return lhs / rhs
    lib = _lib.load()
    order = root_order
    while degree < order // 2:
        w, order = lib.bfs_gl_mul(w, w), order // 2
    lco, rco = lhs.coefficients[:lhs.degree() + 1], rhs.coefficients[:rhs.degree() + 1]
    xoff = _is_genuine_extension(offset)
    as_x = _is_xlist(lco) or _is_xlist(rco) or xoff
    la, ra = _coeff_array(lco, as_x), _coeff_array(rco, as_x)
    if xoff:
        _scale_by_powers(la, offset, la.n)
        _scale_by_powers(ra, offset, ra.n)
    off = 1 if xoff else _base_value(offset)
    lc = _transform(la, la.n, order, w, off, 1)
    rc = _transform(ra, ra.n, order, w, off, 1)
    _inverse_in_place(rc, order)
    _hadamard(lc, rc, order)
    quo = _transform(lc, order, order, lib.bfs_gl_inv(w), 1, lib.bfs_gl_inv(order))
    if xoff:
        _scale_by_powers(quo, offset, order, inverse=True)
    elif as_x:
        _lib.check(lib.bfs_gl_scale(quo.ptr, quo.ptr, order, quo.stride, 3, lib.bfs_gl_inv(off), current_stream()))
    else:
        _lib.check(lib.bfs_gl_scale(quo.ptr, quo.ptr, order, order, 1, lib.bfs_gl_inv(off), current_stream()))
    return Polynomial(quo.to_elements()[:lhs.degree() - rhs.degree() + 1])


# ---- subproduct-tree routines (ntt.py:82-161): zerofier, multi-point evaluation, interpolation over an arbitrary domain.
# Same divide-and-conquer as the reference; every product of degree >= 8 goes through fast_multiply (GPU transforms).
def fast_zerofier(domain, primitive_root, root_order):
    _check_root(primitive_root, root_order)
    if len(domain) == 0:
        return Polynomial([])
    if len(domain) == 1:
        return Polynomial([-domain[0], primitive_root.field.one()])
    half = len(domain) // 2
    return fast_multiply(fast_zerofier(domain[:half], primitive_root, root_order),
                         fast_zerofier(domain[half:], primitive_root, root_order), primitive_root, root_order)


def fast_evaluate(polynomial, domain, primitive_root, root_order):
    _check_root(primitive_root, root_order)
    if len(domain) == 0:
        return []
    if len(domain) == 1:
        return [polynomial.evaluate(domain[0])]
    half = len(domain) // 2
    lz = fast_zerofier(domain[:half], primitive_root, root_order)
    rz = fast_zerofier(domain[half:], primitive_root, root_order)
    return (fast_evaluate(polynomial % lz, domain[:half], primitive_root, root_order) +
            fast_evaluate(polynomial % rz, domain[half:], primitive_root, root_order))


def fast_interpolate(domain, values, primitive_root, root_order):
    _check_root(primitive_root, root_order)
    assert len(domain) == len(values), "cannot interpolate over domain of different length than values list"
    if len(domain) == 0:
        return Polynomial([])
    if len(domain) == 1:
        return Polynomial([values[0]])
    half = len(domain) // 2
    lz = fast_zerofier(domain[:half], primitive_root, root_order)
    rz = fast_zerofier(domain[half:], primitive_root, root_order)
    left_offset = fast_evaluate(rz, domain[:half], primitive_root, root_order)
    right_offset = fast_evaluate(lz, domain[half:], primitive_root, root_order)
    left_targets = [v / d for v, d in zip(values[:half], left_offset)]
    right_targets = [v / d for v, d in zip(values[half:], right_offset)]
    li = fast_interpolate(domain[:half], left_targets, primitive_root, root_order)
    ri = fast_interpolate(domain[half:], right_targets, primitive_root, root_order)
    return li * rz + ri * lz
