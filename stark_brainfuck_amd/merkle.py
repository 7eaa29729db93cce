"""BLAKE2b-512 Merkle trees built on the GPU -- mirror of the reference's `merkle.py`
(/root/reference/code/merkle.py:7-63): `Merkle(data_array)`, `.root()`, `.open(index)`, `Merkle.verify(...)`,
attributes `num_leafs`, `depth`, `leafs`, `nodes`.

Leaf hashing and all inner levels run in HIP kernels (csrc/merkle.hip).  When the leaves are field elements (or an
XArray / BaseArray in HBM) the pickle preimage of every leaf is synthesised on the GPU from the limbs; arbitrary
picklable leaves are pickled on the host, exactly like the reference does, and hashed on the GPU in one batch.
"""
import pickle
from hashlib import blake2b

import numpy as np

from . import _lib
from .algebra import BaseFieldElement
from .arrays import BaseArray, XArray
from .device import DeviceBuffer, current_stream, synchronize
from .extension_field import ExtensionFieldElement
from .ip import NativeTranscript


import contextvars

# verify(): the proof stream being verified can hand out the pickle of an object it holds without walking it (ip.ProofStream.pickle_of)
leaf_pickle_source = contextvars.ContextVar("bfs_leaf_pickle_source", default=None)


def leaf_bytes(element, transcript=None):
    """the reference's pickle.dumps(element): native emitter for this package's element classes (and containers
    of them), CPython's pickle for everything else."""
    source = leaf_pickle_source.get()
    if source is not None and transcript is None:
        b = source(element)
        if b is not None:
            return b
    try:
        return (transcript or NativeTranscript()).dumps(element)
    except TypeError:
        return pickle.dumps(element, protocol=4)


def _tree_shape(n):
    npo2 = 1
    while npo2 < n:
        npo2 <<= 1
    if n == 0:
        npo2 = 0
    return npo2, max(npo2.bit_length() - 1, 0)


def _walk(k):
    while k > 1:
        yield k
        k >>= 1


class Merkle:
    def __init__(self, data_array, _device_nodes=None):
        self.num_leafs = len(data_array)
        self._npo2, self.depth = _tree_shape(self.num_leafs)
        self._data = data_array
        self._leafs = None
        self._nodes_host = None
        self._node_cache = {}
        if _device_nodes is not None:            # tree already built in HBM (Fri.commit)
            self._nodes = _device_nodes
            return
        self._nodes = DeviceBuffer(max(2 * self._npo2, 2) * 8)
        if self.num_leafs:
            self._build(data_array)

    # ---- construction
    def _build(self, data):
        lib, stream = _lib.load(), current_stream()
        n = self.num_leafs
        if isinstance(data, XArray):
            _lib.check(lib.bfs_merkle_build_xfe(data.ptr, data.stride, n, self._nodes.ptr, stream))
        elif isinstance(data, BaseArray):
            _lib.check(lib.bfs_merkle_build_bfe(data.ptr, n, self._nodes.ptr, stream))
        elif all(isinstance(e, ExtensionFieldElement) for e in data):
            arr = XArray.from_elements(data)
            _lib.check(lib.bfs_merkle_build_xfe(arr.ptr, arr.stride, n, self._nodes.ptr, stream))
            synchronize(stream)
        elif all(isinstance(e, BaseFieldElement) for e in data) and len({id(e.field) for e in data}) == 1:
            arr = BaseArray.from_elements(data)
            _lib.check(lib.bfs_merkle_build_bfe(arr.ptr, n, self._nodes.ptr, stream))
            synchronize(stream)
        else:
            self._build_from_bytes([leaf_bytes(e) for e in data])

    def _build_from_bytes(self, preimages):
        lib, stream = _lib.load(), current_stream()
        n = len(preimages)
        lengths = np.fromiter((len(b) for b in preimages), dtype=np.uint32, count=n)
        words = (lengths.astype(np.uint64) + 7) // 8
        offsets = np.zeros(n, dtype=np.uint64)
        np.cumsum(words[:-1], out=offsets[1:])
        total = int(words.sum())
        blob = bytearray(max(total, 1) * 8)
        for b, off in zip(preimages, offsets):
            blob[int(off) * 8:int(off) * 8 + len(b)] = b
        d_data = DeviceBuffer.from_numpy(np.frombuffer(bytes(blob), dtype=np.uint64))
        d_off = DeviceBuffer.from_numpy(offsets)
        d_len = DeviceBuffer.from_numpy(np.frombuffer(np.ascontiguousarray(np.concatenate([lengths, np.zeros(n % 2, np.uint32)])).tobytes(), dtype=np.uint64))
        _lib.check(lib.bfs_merkle_build_bytes(d_data.ptr, d_off.ptr, d_len.ptr, n, self._nodes.ptr, stream))
        synchronize(stream)

    # ---- reference attributes
    @property
    def leafs(self):
        if self._leafs is None:
            d = self._data
            self._leafs = d.to_elements() if isinstance(d, (XArray, BaseArray)) else [leaf for leaf in d]
        return self._leafs

    @property
    def nodes(self):
        """the reference's `nodes` list: 2*npo2 entries, 32 zero bytes where the reference never writes a digest."""
        if self._nodes_host is None:
            npo2, n = self._npo2, self.num_leafs
            raw = self._nodes.to_numpy(2 * npo2 * 8).tobytes() if npo2 else b""
            nodes = [raw[64 * i:64 * i + 64] for i in range(2 * npo2)]
            for i in range(npo2 + n, 2 * npo2):
                nodes[i] = bytes(32)                                     # merkle.py:26
            if npo2:
                nodes[0] = blake2b(bytes(32) + nodes[1]).digest()        # merkle.py:35-41 runs down to index 0
            for k, obj in self._node_cache.items():
                nodes[k] = obj                                           # keep the identity of nodes already handed out
            self._nodes_host = nodes
        return self._nodes_host

    def root(self):
        if self._nodes_host is not None:
            return self._nodes_host[1]
        if getattr(self, "_root", None) is None:
            synchronize()
            self._root = self._nodes.to_numpy(8, offset=8).tobytes()
        return self._root

    def open(self, index):
        """sibling digests from the leaf level up (merkle.py:46-52).  A node is always returned as the SAME bytes
        object, like the reference's `nodes` list: pickle memoises by identity, so this is visible in proof streams."""
        if self.depth == 0:
            return []
        if self._nodes_host is not None:
            nodes = self._nodes_host
            return [nodes[k ^ 1] for k in _walk((1 << self.depth) | index)]
        siblings = [k ^ 1 for k in _walk((1 << self.depth) | index)]
        missing = [sib for sib in siblings if sib not in self._node_cache and sib < self._npo2 + self.num_leafs]
        if missing:              # all digests of the path in one round trip (bfs_gather)
            from .device import gather
            raw = gather([(self._nodes.ptr + 64 * sib, 8, 1) for sib in missing]).tobytes()
            for j, sib in enumerate(missing):
                self._node_cache[sib] = raw[64 * j:64 * j + 64]
        path = []
        for sib in siblings:
            if sib not in self._node_cache:
                self._node_cache[sib] = bytes(32)                 # absent leaf slot: the reference keeps 32 zero bytes there (merkle.py:26)
            path.append(self._node_cache[sib])
        return path

    def prefetch_paths(self, indices, batch):
        """queue the digests that open(i) will need for every i in `indices` on a GatherBatch; returns a function to call after
        batch.run() that stores them (open() then finds everything in the cache)"""
        if self.depth == 0 or self._nodes_host is not None:
            return lambda: None
        wanted, tickets = [], []
        for index in indices:
            for k in _walk((1 << self.depth) | index):
                sib = k ^ 1
                if sib not in self._node_cache and sib not in wanted and sib < self._npo2 + self.num_leafs:
                    wanted.append(sib)
                    tickets.append(batch.add(self._nodes.ptr + 64 * sib, 8, 1))

        def store():
            for sib, ticket in zip(wanted, tickets):
                self._node_cache[sib] = batch.words(ticket).tobytes()
        return store

    @staticmethod
    def verify(root, index, path, element):
        """verifier side (host -- as in the reference, merkle.py:54-63): natively when the objects belong to the proof stream being
        verified (one call, bfs_ps_merkle_verify), else through hashlib."""
        stream = getattr(leaf_pickle_source.get(), "__self__", None)
        if stream is not None and hasattr(stream, "native_path_check"):
            verdict = stream.native_path_check(root, index, None, path, element)
            if verdict is not None:
                return verdict
        running = blake2b(leaf_bytes(element)).digest()
        for node in path:
            running = blake2b(running + node).digest() if index % 2 == 0 else blake2b(node + running).digest()
            index >>= 1
        return running == root
