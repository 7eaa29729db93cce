"""Multi-GPU sharding of independent trace columns / codewords.

The hot path has no data-path collective: column c is transformed and committed entirely on rank c mod G
(`reduce(... table.lde ...)` of the reference, /root/reference/code/brainfuck_stark.py:171-172,194-195, treats columns
independently).  The only exchange is the "final transcript reduction": every rank contributes the 64-byte Merkle
roots of its columns and receives all of them (one all_gather over RCCL/xGMI; `gloo` in the CPU tests).  Field
elements are never summed across ranks -- ncclSum on uint64 would not be reduction mod p.
"""
import contextvars

import numpy as np

# the shared_randomness block this context (thread / task) is inside, if any: its collectives ask it whether a peer has failed
_current_block = contextvars.ContextVar("bfs_shared_randomness_block", default=None)


def check_peers():
    """in front of every collective of a cooperative proof: if a rank of this shared_randomness block has left it through an exception
    (it said so in the process group's store -- out of band, it takes part in no further collective), raise here instead of entering a
    collective that rank will never join.  Outside a block, or without a store: nothing."""
    block = _current_block.get()
    if block is not None:
        block.raise_if_a_peer_failed()


def assign_columns(num_columns, world_size, rank):
    """global column indices owned by `rank`: c mod world_size == rank."""
    assert 0 <= rank < world_size
    return [c for c in range(num_columns) if c % world_size == rank]


def columns_per_rank(num_columns, world_size):
    return [len(assign_columns(num_columns, world_size, r)) for r in range(world_size)]


def gather_roots(local_roots, num_columns, world_size, rank, group=None, device=None, force_collective=False):
    """all-gather of per-column 64-byte roots.  local_roots: {global column index: 64 bytes} for the columns of this
    rank.  Returns the list of all `num_columns` roots, identical on every rank.  device: where the exchanged tensors live
    (a CUDA device for RCCL, None for gloo); force_collective: run the all_gather even for a single rank (tests)."""
    mine = assign_columns(num_columns, world_size, rank)
    assert sorted(local_roots) == mine, "rank %d must supply exactly its own columns %r" % (rank, mine)
    if world_size == 1 and not force_collective:
        return [bytes(local_roots[c]) for c in range(num_columns)]
    import torch
    import torch.distributed as dist
    width = max(columns_per_rank(num_columns, world_size))
    buf = np.zeros((width, 64), dtype=np.uint8)
    for slot, c in enumerate(mine):
        buf[slot] = np.frombuffer(bytes(local_roots[c]), dtype=np.uint8)
    send = torch.from_numpy(buf)
    if device is not None:
        send = send.to(device)
    recv = [torch.empty_like(send) for _ in range(world_size)]
    check_peers()
    dist.all_gather(recv, send, group=group)
    out = [None] * num_columns
    for r in range(world_size):
        rows = recv[r].cpu().numpy()
        for slot, c in enumerate(assign_columns(num_columns, world_size, r)):
            out[c] = rows[slot].tobytes()
    return out


# ---------------------------------------------------------------------------------------------------------------------
# One commitment, several GPUs: SaltedMerkle(list(zip(*codewords))) with the codewords' columns spread over the ranks
# (/root/reference/code/brainfuck_stark.py:178-180, 197-199).  A leaf is the pickle of a whole ROW, so the rows
# have to be brought together: every rank sends, of each of its columns, the slice of rows [s n/G, (s+1) n/G) to rank s (ONE
# all-to-all over RCCL / xGMI: (G-1)/G of the codeword bytes cross the links once), hashes the leaves of its own row range and
# builds the subtree above them; the G subtree roots (64 bytes each) are all-gathered and every rank finishes the top log2(G)
# levels on the host.  Bit-exact with the single-GPU tree: node k of the reference's heap (merkle.py:26-44) at depth log2(G) is
# exactly the root of rank (k - G)'s subtree.  Salts: rank s draws the salts of its rows (they are random); a caller that has to
# reproduce a given salt stream passes `salts` (24 bytes per leaf, leaf order), of which every rank uses its slice.

def row_range(n, world_size, rank):
    assert n % world_size == 0, "the number of rows must be a multiple of the number of ranks"
    m = n // world_size
    return rank * m, m


def exchange_rows(local_columns, planes, n, world_size, rank, group=None, device=None):
    """the all-to-all row transpose.  local_columns: {global column c: int64 tensor (planes[c], n)} for the columns of this rank
    (CPU tensors under gloo, CUDA tensors under RCCL; uint64 values viewed as int64).  planes[c] = 1 for a base column, 3 for an
    extension column (limb planes), for ALL columns.  Returns [tensor (planes[c], n / G) for c in all columns]: this rank's rows.
    device: where the exchange buffers live -- taken from the rank's own columns when it has any; a rank that owns NO column (fewer
    columns than ranks) must say it (under RCCL the empty send buffer has to be a CUDA tensor like everybody else's)."""
    import torch
    import torch.distributed as dist
    num_columns = len(planes)
    assert world_size >= 1 and world_size & (world_size - 1) == 0, "the number of ranks must be a power of two"
    mine = assign_columns(num_columns, world_size, rank)
    assert sorted(local_columns) == mine, "rank %d must supply exactly its own columns %r" % (rank, mine)
    first, m = row_range(n, world_size, rank)
    if world_size == 1:
        return [local_columns[c][:, first:first + m].contiguous() for c in range(num_columns)]
    if mine:
        device = next(iter(local_columns.values())).device if device is None else torch.device(device)
        assert all(t.device == device for t in local_columns.values()), "all columns of a rank live on one device"
    else:
        if device is None:
            assert dist.get_backend(group) != "nccl", "rank %d owns no column: pass device= (RCCL needs CUDA buffers on every rank)" % rank
            device = torch.device("cpu")
        device = torch.device(device)
    # to rank s: for each of my columns (ascending), its planes, rows of s
    send = torch.cat([local_columns[c][:, s * m:(s + 1) * m].reshape(-1) for s in range(world_size) for c in mine]) if mine \
        else torch.empty(0, dtype=torch.int64, device=device)
    my_words = sum(planes[c] for c in mine) * m
    in_splits = [my_words] * world_size
    out_splits = [sum(planes[c] for c in assign_columns(num_columns, world_size, r)) * m for r in range(world_size)]
    recv = torch.empty(sum(out_splits), dtype=torch.int64, device=send.device)
    check_peers()
    dist.all_to_all_single(recv, send.contiguous(), out_splits, in_splits, group=group)
    out, pos = [None] * num_columns, 0
    for r in range(world_size):
        for c in assign_columns(num_columns, world_size, r):
            out[c] = recv[pos:pos + planes[c] * m].reshape(planes[c], m)
            pos += planes[c] * m
    return out


class _DeviceWords:
    """a window of device memory (64-bit words) as an object torch.as_tensor can wrap without copying (__cuda_array_interface__)"""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<i8", "data": (int(ptr), False), "version": 2}


def all_gather_rows(ptr, n, planes, stride, world_size, rank, group=None, device=None, stream=None, force_collective=False):
    """every rank has filled rows [rank n/G, (rank+1) n/G) of `planes` planes of n words (`stride` words apart, first plane at device
    address `ptr`); afterwards every rank holds all rows of all planes.  The one data-path collective of a cooperative proof: the
    combination codeword before FRI (3 planes: 24 n bytes in all, (G-1)/G of them arriving over xGMI).  device given (RCCL): the
    collective runs on the device buffers themselves; device None (gloo in the tests): through host copies."""
    import torch
    import torch.distributed as dist
    from . import _lib
    from .device import current_stream
    first, m = row_range(n, world_size, rank)
    if world_size == 1 and not force_collective:       # (force_collective: the single-rank RCCL test walks the device path)
        return
    lib = _lib.load()
    stream = current_stream() if stream is None else stream
    _lib.check(lib.bfs_stream_synchronize(stream))                 # the kernels that wrote this rank's rows
    check_peers()
    if device is not None:
        for p in range(planes):
            whole = torch.as_tensor(_DeviceWords(ptr + 8 * p * stride, n), device=device)
            mine = whole[first:first + m].clone()                  # (the input of an all-gather must not alias its output)
            dist.all_gather_into_tensor(whole, mine, group=group)
        torch.cuda.synchronize()
        return
    import numpy as np
    for p in range(planes):
        mine = np.empty(m, dtype=np.uint64)
        _lib.check(lib.bfs_memcpy_d2h(mine.ctypes.data, ptr + 8 * (p * stride + first), 8 * m, stream))
        _lib.check(lib.bfs_stream_synchronize(stream))
        send = torch.from_numpy(mine.view(np.int64))
        recv = [torch.empty_like(send) for _ in range(world_size)]
        dist.all_gather(recv, send, group=group)
        whole = np.concatenate([r.numpy() for r in recv]).view(np.uint64)
        _lib.check(lib.bfs_memcpy_h2d(ptr + 8 * p * stride, whole.ctypes.data, 8 * n, stream))
        _lib.check(lib.bfs_stream_synchronize(stream))


def _blake2b_pair(left, right):
    from hashlib import blake2b
    return blake2b(left + right).digest()


class ShardedZippedMerkle:
    """the zipped, salted commitment over row-sharded leaves.

    local_columns / planes / n: see exchange_rows.  build_subtree(rows, first_row, salts) -> an object with .root() and
    .open(i) -> (salt, path) over the m rows this rank received (rows: list over ALL columns of (planes[c], m) tensors); the
    product's builder hashes them on the GPU (gpu_subtree_builder), tests pass a CPU oracle.  salts: None, or 24 n bytes."""

    def __init__(self, local_columns, planes, n, world_size, rank, build_subtree, group=None, salts=None, device=None, rows=None):
        """rows: skip the exchange -- the caller already holds this rank's rows (RowShardedSaltedMerkle)"""
        import torch
        import torch.distributed as dist
        self.n, self.world_size, self.rank, self.group = n, world_size, rank, group
        self.first, self.m = row_range(n, world_size, rank)
        assert world_size & (world_size - 1) == 0, "the top levels are a binary tree over the ranks"
        self.rows = exchange_rows(local_columns, planes, n, world_size, rank, group, device=device) if rows is None else rows
        my_salts = None if salts is None else salts[24 * self.first:24 * (self.first + self.m)]
        self.subtree = build_subtree(self.rows, self.first, my_salts)
        mine = self.subtree.root()
        if world_size == 1:
            roots = [mine]
        else:
            send = torch.frombuffer(bytearray(mine), dtype=torch.uint8)
            if device is not None:
                send = send.to(device)
            recv = [torch.empty_like(send) for _ in range(world_size)]
            check_peers()
            dist.all_gather(recv, send, group=group)
            roots = [r.cpu().numpy().tobytes() for r in recv]
        # heap of the top levels: top[G + r] = subtree root of rank r, top[k] = H(top[2k] || top[2k+1])   (merkle.py:35-41)
        self.top = [None] * (2 * world_size)
        for r in range(world_size):
            self.top[world_size + r] = roots[r]
        for k in range(world_size - 1, 0, -1):
            self.top[k] = _blake2b_pair(self.top[2 * k], self.top[2 * k + 1])

    def root(self):
        return self.top[1]

    def owner(self, index):
        return index // self.m

    def top_path(self, index):
        """the authentication-path nodes above the owner's subtree, bottom up"""
        k, path = self.world_size + self.owner(index), []
        while k > 1:
            path.append(self.top[k ^ 1])
            k >>= 1
        return path

    def open(self, index):
        """(salt, path) of leaf `index` (salted_merkle.py:47-49): a COLLECTIVE call -- the owner of the row supplies the salt and the
        part of the path inside its subtree, every rank appends the top levels."""
        import torch.distributed as dist
        assert 0 <= index < self.n, "leaf index out of range"
        owner = self.owner(index)
        box = [None]
        if self.rank == owner:
            salt, path = self.subtree.open(index - self.first)
            box = [(bytes(salt), [bytes(p) for p in path])]
        if self.world_size > 1:
            check_peers()
            dist.broadcast_object_list(box, src=owner, group=self.group)
        salt, path = box[0]
        return salt, path + self.top_path(index)


def gpu_subtree_builder(ext_flags, make_row=None):
    """build_subtree for ShardedZippedMerkle on the MI355X: the received row slices are hashed by the zipped-row leaf kernel
    (bfs_merkle_build_rows, csrc/rows.hip) where they are -- CUDA tensors -- or after one upload (CPU tensors of a gloo run)."""
    def build(rows, first_row, salts):
        import numpy as np
        from .device import DeviceBuffer
        from .salted_merkle import ZippedSaltedMerkle
        m = rows[0].shape[1]
        keep, columns = [], []
        for t, is_ext in zip(rows, ext_flags):
            if t.is_cuda:
                keep.append(t.contiguous())
                ptr = keep[-1].data_ptr()
            else:
                keep.append(DeviceBuffer.from_numpy(np.ascontiguousarray(t.numpy()).view(np.uint64).reshape(-1)))
                ptr = keep[-1].ptr
            columns.append((ptr, bool(is_ext), 0))
        tree = ZippedSaltedMerkle(columns, m, make_row or (lambda i: None), salts=salts)
        tree._keep_alive = keep
        return tree
    return build


# ---------------------------------------------------------------------------------------------------------------------
# One proof, several GPUs, no data movement: every rank runs the (cheap) polynomial stages of BrainfuckStark.prove on all columns
# and the (expensive: half of a large proof) zipped-row leaf hashing on its own RANGE of rows only; the ranks exchange nothing but
# 64-byte subtree roots and, for the few opened rows, the owner's salt and inner authentication path.  The proof is the single-GPU
# proof, byte for byte, on every rank.  Everything random in the proof must be the same on all ranks: shared_randomness().

class RowShardedSaltedMerkle:
    """SaltedMerkle(list(zip(*codewords))) where every rank holds ALL the columns (device pointers) but hashes only rows
    [rank n/G, (rank+1) n/G).  Same interface as salted_merkle.ZippedSaltedMerkle as far as the prover uses it: root(), leafs[i],
    open(i) (collective), prefetch_salts / prefetch_paths (nothing to prefetch: the owner serves openings)."""

    def __init__(self, columns, n, make_row, world_size, rank, group=None, device=None):
        from .salted_merkle import ZippedSaltedMerkle, _LazyLeafs
        first, m = row_range(n, world_size, rank)
        # extension columns: (pointer to limb plane 0, True, id); the planes are n words apart whatever the range
        local = [(ptr + 8 * first, is_ext, field_id) for ptr, is_ext, field_id in columns]
        sub = ZippedSaltedMerkle(local, m, lambda i: make_row(first + i), limb_stride=n, salt_offset=first, total_rows=n)

        class _Sub:                         # what ShardedZippedMerkle expects of a subtree
            def root(self_inner): return sub.root()
            def open(self_inner, i): return sub.open(i)
        self._sub = sub
        self._tree = ShardedZippedMerkle({}, [], n, world_size, rank, lambda rows, f, s: _Sub(), group=group, device=device, rows=[])
        self.num_leafs, self.depth = n, n.bit_length() - 1
        self._opened, self._salt_objects, self._node_objects = {}, {}, {}
        tree = self

        def salt_of(i):
            return tree.open(i)[0]
        self._leafs = _LazyLeafs(n, make_row, salt_of)

    @property
    def leafs(self):
        return self._leafs

    def root(self):
        return self._tree.root()

    def open(self, index):
        """(salt, path) like ZippedSaltedMerkle.open -- including which objects are SHARED between calls, because the proof is a
        pickle and pickle memoises by identity: one bytes object per tree node and per salt however often it is opened, a new path
        list and a new tuple per call (merkle.py:46-52 on the reference's `nodes` list, salted_merkle.py:47-49)."""
        if index not in self._opened:
            salt, path = self._tree.open(index)                      # collective
            self._opened[index] = (salt, path)
        salt, raw = self._opened[index]
        salt = self._salt_objects.setdefault(index, salt)
        k, path = self.num_leafs + index, []
        for node in raw:
            path.append(self._node_objects.setdefault(k ^ 1, node))  # keyed by heap index of the sibling
            k >>= 1
        return salt, path

    def prefetch_salts(self, indices, batch):
        return lambda: None

    def prefetch_paths(self, indices, batch):
        return lambda: None

    def free(self):
        for name in ("_nodes", "_salts"):
            b = getattr(self._sub, name, None)
            if hasattr(b, "free"):
                b.free()


class _SharedStream:
    """os.urandom replaced by SHAKE-256 of a seed that rank 0 drew and broadcast: every rank of a cooperative proof sees the same bytes
    (randomizers, salts, initials), so the ranks' transcripts cannot diverge.  expand_on_device: bulk randomness (24 bytes of salt
    per leaf) is still expanded on the GPU from 32 bytes of this stream, as with the operating system's urandom."""
    expand_on_device = True

    def __init__(self, seed):
        import hashlib
        self._xof, self._pos, self._buf = hashlib.shake_256(b"bfs-shared-randomness" + seed), 0, b""

    def __call__(self, count):
        end = self._pos + count
        if end > len(self._buf):
            self._buf = self._xof.digest(max(2 * end, 1 << 12))
        out = self._buf[self._pos:end]
        self._pos = end
        return out


class shared_randomness:
    """context manager: inside it the prover modules of THIS context (thread / task) draw their randomness from one stream shared by
    the ranks of `group` (randomness.override: a context variable, so another prover thread of the same process keeps its own source
    and the module-level `urandom` names are left alone).  On a clean exit the ranks compare how far each one read the stream: the
    proofs can only agree if every rank drew the same bytes in the same order, and a divergence here would otherwise show up as
    mismatched collectives later.

    A rank that leaves the block through an exception takes part in NO further collective (its peers may be inside another one: on
    RCCL a mismatched collective hangs or returns garbage -- round-4 advice); it says so in the process group's store instead, under a
    key of this block.  The peers look at that key in front of every collective of the block (check_peers) and while they wait for
    each other at its end, and raise.  A peer that is already inside a collective the failed rank never joins still waits for the
    backend's timeout, as it always did.  Without a store (a process group made from one is rare) the closing gather is the old one."""
    closing_timeout_s = 600.0

    def __init__(self, world_size, rank, group=None, seed=None):
        self.world_size, self.rank, self.group, self.seed = world_size, rank, group, seed
        self._store = None

    def _open_store(self, block_id):
        """the block's corner of the process group's store.  `block_id` is a nonce that rank 0 drew and broadcast in __enter__: the ranks
        agree on it by construction, whatever blocks a rank skipped or entered from other threads before (a per-process serial number,
        as in round 5, drifts apart for the rest of the process as soon as one rank raises in front of a block -- round-5 advice)."""
        try:
            from torch.distributed import distributed_c10d as c10d
            return c10d.PrefixStore("bfs-shared-randomness/%s/" % block_id, c10d._get_default_store())
        except Exception:
            return None

    def raise_if_a_peer_failed(self):
        if self._store is not None and self._store.add("failed_count", 0) > 0:
            failed = [str(r) for r in range(self.world_size) if self._store.check(["failed/%d" % r])]
            raise RuntimeError("rank(s) [%s] left the shared-randomness block with an exception" % ", ".join(failed))

    def _forget_keys(self):
        """a block that ended cleanly leaves nothing behind in the store (rank 0, once every rank is through with the keys)"""
        if self._store is None or self.rank != 0:
            return
        for key in ["arrived", "failed_count", "compared"] + ["failed/%d" % r for r in range(self.world_size)]:
            try:
                self._store.delete_key(key)
            except Exception:       # a store without delete_key (FileStore): the keys are a few bytes under a unique prefix
                return

    def __enter__(self):
        import os
        from . import randomness
        seed = self.seed
        if self.world_size > 1:
            import torch.distributed as dist
            box = [(os.urandom(32) if seed is None else seed, os.urandom(12).hex()) if self.rank == 0 else None]
            dist.broadcast_object_list(box, src=0, group=self.group)
            seed, block_id = box[0] if seed is None else (seed, box[0][1])
            self._store = self._open_store(block_id)
        elif seed is None:
            seed = os.urandom(32)
        self._stream = _SharedStream(seed)
        self._override = randomness.override(self._stream)
        self._override.__enter__()
        self._token = _current_block.set(self)
        return self._stream

    def __exit__(self, exc_type, exc, tb):
        _current_block.reset(self._token)
        self._override.__exit__(exc_type, exc, tb)
        if self.world_size > 1:
            import torch.distributed as dist
            if self._store is not None:
                import time
                if exc_type is not None:
                    try:                            # one key per rank, and a counter the peers poll: two failing ranks do not overwrite each other
                        self._store.set("failed/%d" % self.rank, "1")
                        self._store.add("failed_count", 1)
                    except Exception:               # the process group itself is gone: keep the original error
                        pass
                    return False
                # wait until every rank is here (then the closing gather is matched by construction) or one has failed
                self._store.add("arrived", 1)
                deadline = time.monotonic() + self.closing_timeout_s
                while self._store.add("arrived", 0) < self.world_size:
                    self.raise_if_a_peer_failed()
                    if time.monotonic() > deadline:
                        raise RuntimeError("shared-randomness block: not every rank arrived at its end within %.0f s" % self.closing_timeout_s)
                    time.sleep(1e-4)
                self.raise_if_a_peer_failed()
            # every rank reports (left cleanly?, stream position); without a store a rank that failed still takes part, so that its
            # peers get an error here instead of blocking (round-3 advice)
            reports = [None] * self.world_size
            try:
                dist.all_gather_object(reports, (exc_type is None, self._stream._pos), group=self.group)
            except Exception:                       # the process group itself is gone: nothing to compare, keep the original error
                if exc_type is None:
                    raise
                return False
            if exc_type is None:
                failed = [r for r, (ok, _) in enumerate(reports) if not ok]
                if failed:
                    raise RuntimeError("rank(s) %r left the shared-randomness block with an exception" % (failed,))
                positions = [p for _, p in reports]
                assert len(set(positions)) == 1, "the ranks read the shared random stream to different positions: %r" % (positions,)
                if self._store is not None:
                    try:
                        if self._store.add("compared", 1) >= self.world_size or self.rank == 0:
                            # (rank 0 waits for the others to be past their last read of the block's keys)
                            import time
                            deadline = time.monotonic() + 5.0
                            while self.rank == 0 and self._store.add("compared", 0) < self.world_size and time.monotonic() < deadline:
                                time.sleep(1e-4)
                            self._forget_keys()
                    except Exception:
                        pass
        return False
