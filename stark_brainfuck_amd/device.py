"""Device-memory plumbing for the array API: raw HIP buffers owned through the C ABI, plus adapters for
torch tensors (torch is only used for memory/streams/distributed, never for arithmetic)."""
import ctypes
import weakref

import numpy as np

from . import _lib


_TORCH = []          # [torch module] once a GPU context has been seen (the availability probe costs microseconds per call)


def current_stream():
    """HIP stream handle to enqueue on: torch's current stream when torch has a GPU context, else the default stream."""
    if _TORCH:
        return int(_TORCH[0].cuda.current_stream().cuda_stream)
    try:
        import torch
        if torch.cuda.is_initialized():          # (a flag; is_available() probes the driver on every call)
            _TORCH.append(torch)
            return int(torch.cuda.current_stream().cuda_stream)
    except ImportError:
        pass
    return 0


class DeviceBuffer:
    """`count` uint64 words in HBM, from the library's pool (bfs_malloc_async / bfs_free_async: stream-ordered on the
    stream that is current when the buffer is made, which is also where this package enqueues all its work)."""

    def __init__(self, count):
        self.count = int(count)
        self.nbytes = self.count * 8
        self._stream = current_stream()
        p = ctypes.c_void_p()
        _lib.check(_lib.load().bfs_malloc_async(ctypes.byref(p), max(self.nbytes, 8), self._stream))
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
            stream = current_stream()
            if stream != self._stream:           # used under another stream since: hand the block back only when both are idle
                synchronize(stream)
            _lib.load().bfs_free_async(self.ptr, self._stream)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceView:
    """`count` words starting `offset` words into a DeviceBuffer, which it keeps alive (the codewords of one table inside the buffer
    a batched transform wrote for all tables)"""

    def __init__(self, owner, offset, count):
        self.owner, self.count, self.nbytes = owner, int(count), int(count) * 8
        self.ptr = owner.ptr + 8 * int(offset)

    def to_numpy(self, count=None, offset=0, stream=None):
        count = self.count - offset if count is None else count
        out = np.empty(count, dtype=np.uint64)
        if count:
            _lib.check(_lib.load().bfs_memcpy_d2h(out.ctypes.data, self.ptr + 8 * offset, count * 8,
                                                  stream if stream is not None else current_stream()))
        return out

    def free(self):
        """drop the share of the underlying buffer (which goes back to the pool with its last view)"""
        self.owner, self.ptr = None, None


class GatherBatch:
    """scattered reads collected first and fetched in ONE round trip: add() returns a ticket, run() gathers, words(ticket) gives
    that request's words.  (A proof opens ~10 rows of three trees; one gather per row and path costs ~30 us each.)"""

    def __init__(self):
        self._requests, self._spans, self._out = [], [], None

    def add(self, ptr, nwords, stride):
        start = self._spans[-1][1] if self._spans else 0
        self._requests.append((ptr, nwords, stride))
        self._spans.append((start, start + nwords))
        return len(self._spans) - 1

    def run(self, stream=None):
        self._out = gather(self._requests, stream)

    def words(self, ticket):
        lo, hi = self._spans[ticket]
        return self._out[lo:hi]


def pinned_empty(shape):
    """uninitialised uint64 numpy array in pinned host memory from the library's pool (bfs_host_alloc); the memory goes
    back to the pool when the array and every view of it are gone.  H2D copies from it run at link speed."""
    shape = (int(shape),) if np.isscalar(shape) else tuple(int(d) for d in shape)
    count = int(np.prod(shape, dtype=np.int64)) if shape else 1
    lib = _lib.load()
    p = ctypes.c_void_p()
    _lib.check(lib.bfs_host_alloc(ctypes.byref(p), max(count * 8, 8)))
    raw = (ctypes.c_uint64 * max(count, 1)).from_address(p.value)
    weakref.finalize(raw, lib.bfs_host_free, p.value)
    return np.ctypeslib.as_array(raw)[:count].reshape(shape)


def pool_stats():
    """(bytes handed out, bytes cached) of the library's HBM pool"""
    live, cached = ctypes.c_size_t(), ctypes.c_size_t()
    _lib.check(_lib.load().bfs_pool_stats(ctypes.byref(live), ctypes.byref(cached)))
    return live.value, cached.value


def pool_trim():
    """give the cached HBM blocks back to the driver"""
    _lib.check(_lib.load().bfs_pool_trim())


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
