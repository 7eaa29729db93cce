"""Salted Merkle tree -- mirror of the reference's `salted_merkle.py` (/root/reference/code/salted_merkle.py:7-68).
Leaf i is `(element, salt_i)` with a 24-byte random salt; its digest is blake2b(pickle(element) + pickle(salt)),
two separate pickles concatenated (:32-35).  Preimages are assembled on the host, hashed on the GPU.
"""
import pickle
from hashlib import blake2b
from os import urandom          # module-level name on purpose: callers patch `salted_merkle.urandom` for determinism
from .randomness import source as random_source

import ctypes

from . import _lib
from .device import DeviceBuffer, current_stream
from .ip import NativeTranscript
from .merkle import Merkle, leaf_bytes


class SaltedMerkle(Merkle):
    def __init__(self, data_array):
        n = len(data_array)
        assert n & (n - 1) == 0 and n > 0, \
            f"in SaltedMerkle.__init__, next_power_of_two = {n} =/= 1 << self.depth"
        draw = random_source(urandom)
        leafs = [(element, draw(24)) for element in data_array]
        t = NativeTranscript()
        preimages = [leaf_bytes(e, t) + pickle.dumps(s, protocol=4) for e, s in leafs]
        Merkle.__init__(self, preimages, _device_nodes=None)
        self._leafs = leafs

    def _build(self, data):
        self._build_from_bytes(data)

    def open(self, index):
        return (self._leafs[index][1], Merkle.open(self, index))

    @staticmethod
    def verify(root, index, salt, path, element):
        from .merkle import leaf_pickle_source
        stream = getattr(leaf_pickle_source.get(), "__self__", None)
        if stream is not None and hasattr(stream, "native_path_check"):
            verdict = stream.native_path_check(root, index, salt, path, element)      # one native call for objects of the stream being verified
            if verdict is not None:
                return verdict
        running = blake2b(leaf_bytes(element) + pickle.dumps(salt, protocol=4)).digest()
        for node in path:
            running = blake2b(running + node).digest() if index % 2 == 0 else blake2b(node + running).digest()
            index >>= 1
        return running == root


class _LazyLeafs:
    """`leafs` of a ZippedSaltedMerkle: (row tuple, salt) pairs made on first access and then kept, so that a row opened
    twice is the same Python object both times (pickle memoises by identity)."""

    def __init__(self, n, make_row, salt_of):
        self._n, self._make_row, self._salt_of, self._cache = n, make_row, salt_of, {}

    def __len__(self):
        return self._n

    def __getitem__(self, index):
        if index not in self._cache:
            self._cache[index] = (self._make_row(index), self._salt_of(index))
        return self._cache[index]


class ZippedSaltedMerkle(SaltedMerkle):
    """SaltedMerkle(list(zip(*codewords))) for codewords that live in HBM (brainfuck_stark.py:178-179, 197-198).
    columns: list of (device pointer, is_extension, base_field_id); make_row(i) builds the tuple of element objects of
    row i on demand (only opened rows are ever materialised).  The pickle of every row is synthesised and hashed on the
    GPU (bfs_merkle_build_rows, csrc/rows.hip).

    Salts: when `salted_merkle.urandom` is the operating system's (the normal case) they are expanded on the GPU from 32
    bytes of it and never visit the host (bfs_random_fill); when a test has replaced `urandom` to reproduce the reference's
    byte stream they are drawn from it, 24 bytes per leaf in leaf order (salted_merkle.py:25); `salts` (24 n bytes) supplies them
    directly (the row-sharded commitment of shard.py hands every rank its slice of one stream)."""

    def __init__(self, columns, n, make_row, salts=None, limb_stride=None, salt_offset=0, total_rows=None):
        """limb_stride / salt_offset / total_rows: the rows are the range [salt_offset, salt_offset + n) of columns of total_rows
        elements (a rank of a row-sharded commitment, shard.RowShardedSaltedMerkle): `columns` point at the first row of the range,
        extension limb planes are limb_stride words apart, and the salts are the range's slice of the stream for ALL rows."""
        import os
        assert n & (n - 1) == 0 and n > 0, f"in SaltedMerkle.__init__, next_power_of_two = {n} =/= 1 << self.depth"
        lib, stream = _lib.load(), current_stream()
        limb_stride = n if limb_stride is None else int(limb_stride)
        total_rows = n if total_rows is None else int(total_rows)
        self.num_leafs = n
        self._npo2, self.depth = n, n.bit_length() - 1
        self._data = None
        self._nodes_host = None
        self._node_cache = {}
        self._nodes = DeviceBuffer(2 * n * 8)
        cols = (_lib.RowColumn * len(columns))()
        for c, (ptr, is_ext, field_id) in zip(cols, columns):
            c.d_values, c.is_ext, c.field_id = ptr, int(is_ext), field_id
        draw = random_source(urandom)
        if salts is None and (draw is os.urandom or getattr(draw, "expand_on_device", False)):
            # one stream for all total_rows leaves, expanded from 32 bytes (every rank of a sharded commitment expands the same one)
            words = (3 * total_rows + 7) // 8 * 8
            self._salts = DeviceBuffer(words)
            _lib.check(lib.bfs_random_fill(draw(32), self._salts.ptr, words, stream))
            first = self._salts.ptr + 24 * salt_offset
            root = ctypes.create_string_buffer(64)
            _lib.check(lib.bfs_merkle_build_rows_root(cols, len(columns), n, limb_stride, first, 1, self._nodes.ptr, root, stream))
            self._root = root.raw              # (came back with the call's last read-back)

            cache = self._salt_cache = {}
            d_salts = self._salts           # the closure must not hold `self`: tree -> leafs -> closure -> tree would be a cycle
            self._salt_base = first

            def salt_of(i):
                from .device import gather
                if i not in cache:
                    cache[i] = gather([(d_salts.ptr + 24 * (salt_offset + i), 3, 1)]).tobytes()
                return cache[i]
        else:
            if salts is None:
                salts = draw(24 * total_rows)[24 * salt_offset:24 * (salt_offset + n)]   # the same bytes as one urandom(24) per leaf
            assert len(salts) == 24 * n, "24 bytes of salt per leaf"
            keep = ctypes.create_string_buffer(salts, len(salts))
            self._salt_host = keep                   # bfs_stark_push_openings reads opened salts from here
            root = ctypes.create_string_buffer(64)
            _lib.check(lib.bfs_merkle_build_rows_root(cols, len(columns), n, limb_stride, ctypes.addressof(keep), 0,        # (not ctypes.cast: it makes `keep` part of a reference cycle)
                                                      self._nodes.ptr, root, stream))
            self._root = root.raw

            def salt_of(i):
                return salts[24 * i:24 * i + 24]
        self._leafs = _LazyLeafs(n, make_row, salt_of)

    def prefetch_salts(self, indices, batch):
        """queue the salts of rows `indices` on a GatherBatch (device-made salts only); returns the function that stores them"""
        cache = getattr(self, "_salt_cache", None)
        if cache is None:
            return lambda: None
        wanted = [i for i in dict.fromkeys(indices) if i not in cache]
        tickets = [batch.add(self._salt_base + 24 * i, 3, 1) for i in wanted]

        def store():
            for i, ticket in zip(wanted, tickets):
                cache[i] = batch.words(ticket).tobytes()
        return store
