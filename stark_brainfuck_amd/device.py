"""Device-memory plumbing for the array API: raw HIP buffers owned through the C ABI, plus adapters for
torch tensors (torch is only used for memory/streams/distributed, never for arithmetic)."""
import ctypes

import numpy as np

from . import _lib


def current_stream():
    """HIP stream handle to enqueue on: torch's current stream when torch has a GPU context, else the default stream."""
    try:
        import torch
        if torch.cuda.is_available() and torch.cuda.is_initialized():
            return int(torch.cuda.current_stream().cuda_stream)
    except ImportError:
        pass
    return 0


class DeviceBuffer:
    """`count` uint64 words in HBM (hipMalloc through bfs_malloc)."""

    def __init__(self, count):
        self.count = int(count)
        self.nbytes = self.count * 8
        p = ctypes.c_void_p()
        _lib.check(_lib.load().bfs_malloc(ctypes.byref(p), max(self.nbytes, 8)))
        self.ptr = p.value

    @classmethod
    def from_numpy(cls, a, stream=None):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        buf = cls(a.size)
        if a.size:
            _lib.check(_lib.load().bfs_memcpy_h2d(buf.ptr, a.ctypes.data, a.nbytes, stream if stream is not None else current_stream()))
        return buf

    def to_numpy(self, count=None, offset=0, stream=None):
        count = self.count - offset if count is None else count
        out = np.empty(count, dtype=np.uint64)
        if count:
            _lib.check(_lib.load().bfs_memcpy_d2h(out.ctypes.data, self.ptr + 8 * offset, count * 8,
                                                  stream if stream is not None else current_stream()))
        return out

    def free(self):
        if self.ptr:
            _lib.load().bfs_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def device_ptr(x):
    """device pointer of a DeviceBuffer, a torch CUDA tensor of 64-bit integers, or a raw int."""
    if isinstance(x, int):
        return x
    if hasattr(x, "ptr") and not hasattr(x, "data_ptr"):
        return x.ptr          # DeviceBuffer, or a borrowed view into a native allocation
    if hasattr(x, "data_ptr") and hasattr(x, "is_cuda"):
        if not x.is_cuda:
            raise ValueError("expected a tensor in HBM (cuda device)")
        if x.element_size() != 8 or not x.is_contiguous():
            raise ValueError("expected a contiguous tensor of 64-bit integers")
        return x.data_ptr()
    raise TypeError("not a device array: %r" % type(x))


def synchronize(stream=None):
    _lib.check(_lib.load().bfs_stream_synchronize(stream if stream is not None else current_stream()))


def gather(requests, stream=None):
    """scattered reads from HBM in one round trip (bfs_gather): requests = [(device address, nwords, stride in words), ...];
    returns a numpy uint64 array with the words of all requests, in order."""
    lib = _lib.load()
    reqs = (_lib.GatherRequest * len(requests))()
    total = 0
    for r, (ptr, nwords, stride) in zip(reqs, requests):
        r.d_base, r.nwords, r.stride = ptr, nwords, stride
        total += nwords
    out = np.empty(total, dtype=np.uint64)
    if requests:
        _lib.check(lib.bfs_gather(reqs, len(requests), out.ctypes.data, stream if stream is not None else current_stream()))
    return out
