"""Multi-GPU sharding of independent trace columns / codewords.

The hot path has no data-path collective: column c is transformed and committed entirely on rank c mod G
(`reduce(... table.lde ...)` of the reference, /root/reference/code/brainfuck_stark.py:171-172,194-195, treats columns
independently).  The only exchange is the "final transcript reduction": every rank contributes the 64-byte Merkle
roots of its columns and receives all of them (one all_gather over RCCL/xGMI; `gloo` in the CPU tests).  Field
elements are never summed across ranks -- ncclSum on uint64 would not be reduction mod p.
"""
import numpy as np


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
    dist.all_gather(recv, send, group=group)
    out = [None] * num_columns
    for r in range(world_size):
        rows = recv[r].cpu().numpy()
        for slot, c in enumerate(assign_columns(num_columns, world_size, r)):
            out[c] = rows[slot].tobytes()
    return out
