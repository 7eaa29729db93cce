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


# ---------------------------------------------------------------------------------------------------------------------
# one commitment over several ranks (shard.ShardedZippedMerkle): the base commitment of the reference's own proofs, rebuilt
# under world-size-2 and -4 gloo with the CPU oracle doing each rank's hashing, must give the reference's root

def _base_codewords_by_the_oracle(name):
    """the 17 columns of the reference's first commitment (brainfuck_stark.py:162-180) for golden `name`, recomputed on the CPU:
    the product's host code pads the trace and samples the randomizers from the golden's byte stream, the oracle interpolates
    (INTT over the omicron subgroup + the rank-one randomizer correction) and evaluates on the FRI coset.  Returns
    (golden, planes, [uint64 array (planes, n)], salts)."""
    import hashlib
    import json
    import sys
    sys.path.insert(0, ROOT)
    from oracle import ref_oracle as o
    from stark_brainfuck_amd.brainfuck_stark import BrainfuckStark
    from stark_brainfuck_amd.table import sample_base, sample_ext_many
    from stark_brainfuck_amd.vm import VirtualMachine
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "stark_%s.json" % name)))
    buf = hashlib.shake_256(b"bfs-golden-urandom" + name.encode()).digest(g["urandom_bytes"] + 64)
    pos = [0]

    def urandom(k):
        pos[0] += k
        return buf[pos[0] - k:pos[0]]
    P = o.P
    program = VirtualMachine.compile(g["program"])
    running_time, inputs, outputs = VirtualMachine.run(program, input_data=list(g["input"]))
    pm, mm, im, inm, om = VirtualMachine.simulate(program, input_data=list(inputs))
    stark = BrainfuckStark(running_time, len(mm), program, inputs, outputs)
    for table, matrix in zip(stark.tables, (pm, im, mm, inm, om)):
        table.matrix = matrix
    for table in (stark.processor_table, stark.memory_table, stark.instruction_table, stark.input_table, stark.output_table):
        table.pad()
    n = stark.fri.domain.length
    offset, omega = stark.fri.domain.offset.value, stark.fri.domain.omega.value
    count = stark.max_degree + 1
    rc = sample_ext_many(urandom(27 * count), count, 9)                              # randomizer polynomial (:162-165)
    columns, planes = [np.stack([o.fast_coset_evaluate(np.ascontiguousarray(rc[k]), offset, omega, n) for k in range(3)])], [3]
    for t in stark.tables:
        h = t.height
        cols = t.base_array().reshape(t.base_width, h) if h else None
        rand = [sample_base(urandom(24)) for _ in range(t.base_width)] if h and t.num_randomizers else None
        for c in range(t.base_width):
            if h == 0:
                columns.append(np.zeros((1, n), dtype=np.uint64))
            else:
                f0 = o.intt(t.omicron.value, np.ascontiguousarray(cols[c]))
                coeffs = np.zeros(h + 1, dtype=np.uint64)
                coeffs[:h] = f0
                if rand is not None:
                    acc = 0
                    for v in f0[::-1]:
                        acc = (acc * omega + int(v)) % P                              # f0(omega)
                    cc = (rand[c] - acc) * pow((pow(omega, h, P) - 1) % P, P - 2, P) % P
                    coeffs[0] = (int(coeffs[0]) - cc) % P                             # f = f0 + c (X^h - 1)
                    coeffs[h] = cc
                columns.append(o.fast_coset_evaluate(coeffs, offset, omega, n).reshape(1, n))
            planes.append(1)
    salts = urandom(24 * n)
    return g, planes, columns, salts


def _sharded_worker(rank, world, port, name, q):
    import hashlib
    import sys
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle import ref_oracle as o
    from stark_brainfuck_amd import shard
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g, planes, columns, salts = _base_codewords_by_the_oracle(name)
    n = columns[0].shape[1]
    # the reconstruction is the reference's: per-column digests of the golden
    for c, col in enumerate(columns):
        limbs = np.ascontiguousarray(col.T, dtype="<u8")
        assert hashlib.sha256(limbs.tobytes()).hexdigest() == g["base_tree"]["columns_sha"][c], c
    mine = shard.assign_columns(len(planes), world, rank)
    local = {c: torch.from_numpy(columns[c].view(np.int64).copy()) for c in mine}

    class OracleSubtree:                                  # salted_merkle.py:25-35 + merkle.py:26-52 on this rank's rows
        def __init__(self, rows, first, my_salts):
            m = rows[0].shape[1]
            self.salts = [my_salts[24 * i:24 * i + 24] for i in range(m)]
            leaves = []
            for i in range(m):
                items = []
                for t in rows:
                    v = t[:, i].numpy().view(np.uint64)
                    items.append(o.make_xfe([int(x) for x in v]) if len(v) == 3 else o.make_bfe(int(v[0])))
                leaves.append(o.salted_leaf_bytes(tuple(items), self.salts[i]))
            self.tree = o.MerkleOracle(leaves)

        def root(self):
            return self.tree.root()

        def open(self, i):
            return self.salts[i], self.tree.open(i)
    tree = shard.ShardedZippedMerkle(local, planes, n, world, rank, OracleSubtree, salts=salts)
    # openings are collective; every rank gets the full path of every leaf
    paths = {i: tree.open(i) for i in (0, 1, n // 2 - 1, n // 2, n - 1)}
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, tree.root().hex(), g["base_tree"]["root"], {i: (s.hex(), [p.hex() for p in path]) for i, (s, path) in paths.items()},
           g["base_tree"]["salt0"], g["base_tree"]["salt_last"]))


@pytest.mark.parametrize("name,world", [("plus1", 2), ("loop", 2), ("loop", 4)])
def test_row_sharded_commitment_reproduces_the_reference_root(name, world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    import hashlib
    for rank, root, want, paths, salt0, salt_last in got:
        assert root == want, "rank %d" % rank
        assert paths == got[0][3]
        assert paths[0][0] == salt0 and paths[max(paths)][0] == salt_last
    # the paths are the single tree's: they lead from the leaf digest ... to the root; check the lengths and the top node
    some = got[0][3]
    depth = len(some[0][1])
    assert all(len(p[1]) == depth for p in some.values())


def test_exchange_rows_is_a_transpose_for_one_rank():
    import torch
    from stark_brainfuck_amd import shard
    cols = {c: torch.arange(3 * 8 if c == 0 else 8, dtype=torch.int64).reshape(3 if c == 0 else 1, 8) + 100 * c for c in range(3)}
    out = shard.exchange_rows(cols, [3, 1, 1], 8, 1, 0)
    assert all((out[c] == cols[c]).all() for c in range(3))
    assert shard.row_range(16, 4, 3) == (12, 4)


def _few_columns_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from stark_brainfuck_amd import shard
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    planes, n = [3, 1, 1], 16
    full = {c: (torch.arange(planes[c] * n, dtype=torch.int64).reshape(planes[c], n) + 1000 * c) for c in range(3)}
    mine = shard.assign_columns(3, world, rank)
    rows = shard.exchange_rows({c: full[c] for c in mine}, planes, n, world, rank)     # rank 3 owns nothing and still takes part
    first, m = shard.row_range(n, world, rank)
    ok = all((rows[c] == full[c][:, first:first + m]).all() for c in range(3))
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, bool(ok), len(mine)))


def test_exchange_rows_with_a_rank_that_owns_no_column():
    """fewer columns than ranks: the empty-handed rank's buffers live on the collective's device and the transpose is
    still exact"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_few_columns_worker, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [g[1] for g in got] == [True] * 4 and [g[2] for g in got] == [1, 1, 1, 0]


def _failing_rank_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from stark_brainfuck_amd import shard
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    outcome = "clean"
    try:
        with shard.shared_randomness(world, rank) as stream:
            stream(16)
            if rank == 1:
                raise ValueError("rank 1 fails inside the block")
    except ValueError:
        outcome = "own error"
    except RuntimeError as e:
        outcome = "told: " + str(e)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, outcome))


def _failing_rank_mid_block_worker(rank, world, port, q):
    import sys
    import time
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from stark_brainfuck_amd import shard
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    outcome, entered = "clean", False
    try:
        with shard.shared_randomness(world, rank) as stream:
            stream(16)
            if rank == 1:
                raise ValueError("rank 1 fails before the block's collective")
            time.sleep(1.0)                  # rank 1 has long left; this rank now reaches a collective of the block
            shard.check_peers()
            entered = True                   # (not reached: the all_gather below would wait for a rank that never comes)
            dist.all_gather([torch.zeros(1) for _ in range(world)], torch.zeros(1))
    except ValueError:
        outcome = "own error"
    except RuntimeError as e:
        outcome = "told: " + str(e)
    # the process group is intact: no rank is stuck in, or has skipped, a collective
    dist.barrier()
    with shard.shared_randomness(world, rank) as stream:      # and the next block of the same group starts from a clean slate
        stream(8)
    dist.destroy_process_group()
    q.put((rank, outcome, entered))


def test_a_failing_rank_issues_no_collective_and_its_peers_find_out_in_front_of_theirs():
    """round-4 advice: a rank that leaves the shared-randomness block through an exception must not issue the closing gather while
    its peers may be inside another collective of the block (on RCCL a mismatched collective); it flags the failure in the process
    group's store, and a peer raises in front of its next collective of the block (shard.check_peers) instead of entering it"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_failing_rank_mid_block_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {r: (o, e) for r, o, e in (q.get(timeout=120) for _ in procs)}
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert got[1] == ("own error", False)
    assert got[0][0].startswith("told:") and "[1]" in got[0][0] and got[0][1] is False, got


def test_a_failing_rank_in_shared_randomness_becomes_an_error_on_the_others():
    """a rank that leaves the shared-randomness block through an exception turns into an error on the ranks that left it cleanly,
    which raise at the end of the block instead of blocking there (round-3 advice; since round 5 through the store, not a gather)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_failing_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert got[1] == "own error"
    assert got[0].startswith("told:") and "[1]" in got[0], got
