"""HBM-resident vectors of field elements: the array side of the API.

The reference's functions take Python lists of element objects.  This package accepts those too (for
drop-in use and parity tests), but converting 2^24 Python objects costs seconds, so the same functions also accept
and return these array types, which never leave the GPU:

    BaseArray   n base-field elements, one uint64 each
    XArray      n extension-field elements, limb-major: c0[0..n) c1[0..n) c2[0..n)   (3 uint64 planes)
"""
import numpy as np

from . import _lib
from .algebra import BaseField, BaseFieldElement
from .device import DeviceBuffer, current_stream, device_ptr, synchronize
from .extension_field import ExtensionField, ExtensionFieldElement
from .univariate import Polynomial

try:                                    # list <-> buffer conversions in C (cpyext/fastlist.c, built by stark_brainfuck_amd.build):
    from . import _fastlist             # ~40 ns per element instead of ~1 us.  Host plumbing only; the Python loops below give
except ImportError:                     # the same objects when the extension has not been built
    _fastlist = None


class BaseArray:
    def __init__(self, buf, n, field=None, batch=1):
        self.buf, self.n, self.batch = buf, int(n), int(batch)
        self.field = field if field is not None else BaseField.main()

    @property
    def ptr(self):
        return device_ptr(self.buf)

    def __len__(self):
        return self.n

    @classmethod
    def from_numpy(cls, a, field=None):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        if a.ndim == 2:
            return cls(DeviceBuffer.from_numpy(a.reshape(-1)), a.shape[1], field, batch=a.shape[0])
        return cls(DeviceBuffer.from_numpy(a), a.size, field)

    @classmethod
    def empty(cls, n, field=None, batch=1):
        return cls(DeviceBuffer(int(n) * int(batch)), n, field, batch)

    @classmethod
    def from_elements(cls, elements):
        field = elements[0].field if len(elements) else BaseField.main()
        if _fastlist is not None:
            out = np.empty(len(elements), dtype=np.uint64)
            try:
                _fastlist.pack_base(elements, out)
                return cls.from_numpy(out, field)
            except OverflowError:       # a value outside [0, 2^64): the Python path below reports it
                pass
        return cls.from_numpy(np.fromiter((e.value for e in elements), dtype=np.uint64, count=len(elements)), field)

    def to_numpy(self):
        synchronize()
        a = self.buf.to_numpy(self.n * self.batch)
        return a.reshape(self.batch, self.n) if self.batch > 1 else a

    def to_elements(self):
        f = self.field
        if _fastlist is not None:
            return _fastlist.unpack_base(np.ascontiguousarray(self.to_numpy().reshape(-1)), BaseFieldElement, f)
        return [BaseFieldElement(int(v), f) for v in self.to_numpy().reshape(-1)]


class XArray:
    def __init__(self, buf, n, field=None, stride=None):
        self.buf, self.n = buf, int(n)
        self.stride = int(stride) if stride is not None else self.n
        self.field = field if field is not None else ExtensionField.main()

    @property
    def ptr(self):
        return device_ptr(self.buf)

    def __len__(self):
        return self.n

    @classmethod
    def from_numpy(cls, soa, field=None):
        soa = np.ascontiguousarray(soa, dtype=np.uint64)
        assert soa.ndim == 2 and soa.shape[0] == 3, "expected limb-major array of shape (3, n)"
        return cls(DeviceBuffer.from_numpy(soa.reshape(-1)), soa.shape[1], field)

    @classmethod
    def empty(cls, n, field=None):
        return cls(DeviceBuffer(3 * int(n)), n, field)

    @classmethod
    def from_elements(cls, elements):
        field = elements[0].field if len(elements) else ExtensionField.main()
        if _fastlist is not None:
            out = np.empty((3, len(elements)), dtype=np.uint64)
            try:
                _fastlist.pack_ext(elements, out)
                return cls.from_numpy(out, field)
            except OverflowError:
                pass
        soa = np.zeros((3, len(elements)), dtype=np.uint64)
        for i, e in enumerate(elements):
            for k, c in enumerate(e.polynomial.coefficients):
                soa[k, i] = c.value
        return cls.from_numpy(soa, field)

    def to_numpy(self):
        synchronize()
        if self.stride == self.n:
            return self.buf.to_numpy(3 * self.n).reshape(3, self.n)
        return np.stack([self.buf.to_numpy(self.n, offset=k * self.stride) for k in range(3)])

    def to_elements(self):
        soa = self.to_numpy()
        f = self.field
        if _fastlist is not None:
            return _fastlist.unpack_ext(np.ascontiguousarray(soa), ExtensionFieldElement, Polynomial, BaseFieldElement, f, f._base())
        return [f.from_limbs([int(soa[0, i]), int(soa[1, i]), int(soa[2, i])]) for i in range(self.n)]


def raw_ntt(src_ptr, n_in, in_stride, dst_ptr, out_stride, log_n, batch, root, shift=1, post_scale=1, stream=None):
    """thin call into bfs_gl_ntt (include/bfstark.h)."""
    _lib.check(_lib.load().bfs_gl_ntt(src_ptr, n_in, in_stride, dst_ptr, out_stride, log_n, batch, root, shift, post_scale,
                                      stream if stream is not None else current_stream()))
