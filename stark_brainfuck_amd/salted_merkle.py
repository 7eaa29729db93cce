"""Salted Merkle tree -- mirror of the reference's `salted_merkle.py` (/root/reference/code/salted_merkle.py:7-68).
Leaf i is `(element, salt_i)` with a 24-byte random salt; its digest is blake2b(pickle(element) + pickle(salt)),
two separate pickles concatenated (:32-35).  Preimages are assembled on the host, hashed on the GPU.
"""
import pickle
from hashlib import blake2b
from os import urandom          # module-level name on purpose: callers patch `salted_merkle.urandom` for determinism

from .ip import NativeTranscript
from .merkle import Merkle, leaf_bytes


class SaltedMerkle(Merkle):
    def __init__(self, data_array):
        n = len(data_array)
        assert n & (n - 1) == 0 and n > 0, \
            f"in SaltedMerkle.__init__, next_power_of_two = {n} =/= 1 << self.depth"
        leafs = [(element, urandom(24)) for element in data_array]
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
        running = blake2b(leaf_bytes(element) + pickle.dumps(salt, protocol=4)).digest()
        for node in path:
            running = blake2b(running + node).digest() if index % 2 == 0 else blake2b(node + running).digest()
            index >>= 1
        return running == root
