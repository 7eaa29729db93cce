"""world_size-2 `gloo` test of the multi-GPU path's host logic: column sharding and the all-gather of per-column
roots (the only collective of the design).  The per-column "work" here is the CPU oracle, standing in for the GPU."""
import hashlib
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _column_root(c):
    import sys
    sys.path.insert(0, ROOT)
    from oracle import ref_oracle as o
    n = 256
    v = o.felt_array(0x5EED + (c << 32), 0, n)
    out = o.ntt(o.primitive_nth_root(n), v)
    return hashlib.blake2b(np.ascontiguousarray(out).tobytes()).digest()


def _worker(rank, world, port, ncols, q):
    import sys
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from stark_brainfuck_amd.shard import assign_columns, gather_roots
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = assign_columns(ncols, world, rank)
    roots = gather_roots({c: _column_root(c) for c in mine}, ncols, world, rank)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, mine, [r.hex() for r in roots]))


@pytest.mark.parametrize("ncols", [8, 5])
def test_two_rank_shard_and_gather(ncols):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ncols, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    expected = [_column_root(c).hex() for c in range(ncols)]
    owned = []
    for rank, mine, roots in got:
        assert roots == expected
        assert all(c % 2 == rank for c in mine)
        owned += mine
    assert sorted(owned) == list(range(ncols))


def test_assignment_is_a_partition():
    from stark_brainfuck_amd.shard import assign_columns, columns_per_rank, gather_roots
    for world in (1, 2, 4, 8):
        for ncols in (1, 8, 26):
            cols = sorted(c for r in range(world) for c in assign_columns(ncols, world, r))
            assert cols == list(range(ncols))
            assert sum(columns_per_rank(ncols, world)) == ncols
    assert gather_roots({0: b"\1" * 64, 1: b"\2" * 64}, 2, 1, 0) == [b"\1" * 64, b"\2" * 64]
