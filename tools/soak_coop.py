"""Randomised soak of cooperative proving (BrainfuckStark.cooperate): `world` ranks sharing this GPU and exchanging over gloo prove
random Brainfuck programs together; every rank's proof must be byte-identical to the proof ONE prover writes from the same
randomness, and verify() must accept it.     usage: python tools/soak_coop.py [seconds] [seed] [world]"""
import hashlib
import os
import socket
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def worker(rank, world, port, budget, seed, q):
    import numpy as np
    import torch.distributed as dist
    from soak_stark import Stream, random_program
    from stark_brainfuck_amd import brainfuck_stark, salted_merkle, table
    from stark_brainfuck_amd.brainfuck_stark import BrainfuckStark
    from stark_brainfuck_amd.vm import VirtualMachine
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(seed)              # the same programs on every rank
    t0, count, domains = time.time(), 0, set()
    while True:
        flag = [time.time() - t0 < budget]
        dist.broadcast_object_list(flag, src=0)    # rank 0's clock decides when to stop
        if not flag[0]:
            break
        code = random_program(rng)
        inp = [chr(int(c)) for c in rng.integers(1, 127, code.count(",") + 3)]
        program = VirtualMachine.compile(code)
        try:
            matrices = VirtualMachine.simulate(program, input_data=inp, max_cycles=20000)
        except AssertionError:
            continue
        if len(matrices[4]) and ((matrices[4]._ids == 1).any() or int(matrices[4].values.max()) >= 256):
            continue
        if int(matrices[0].values[:, 4].max()) >> 32:
            continue
        rt = len(matrices[0])
        inputs = inp[:len(matrices[3])]
        outputs = [chr(int(v) % 256) for v in matrices[4].values.reshape(-1)]
        proofs = []
        for coop in (True, False):
            stream = Stream(code.encode())
            for mod in (brainfuck_stark, salted_merkle, table):
                mod.urandom = stream
            stark = BrainfuckStark(rt, len(matrices[1]), program, inputs, outputs)
            if stark.fri.domain.length < 4 * world:
                break
            if coop:
                stark.cooperate(world, rank)
            proofs.append(stark.prove(program, *matrices))
        if len(proofs) < 2:
            continue
        assert proofs[0] == proofs[1], "rank %d: the cooperative proof differs from the single-prover proof for %r" % (rank, code)
        if rank == 0:
            assert BrainfuckStark(rt, len(matrices[1]), program, inputs, outputs).verify(proofs[0]) is True
        count += 1
        domains.add(stark.fri.domain.length)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, count, sorted(domains)))


def main():
    import torch.multiprocessing as mp
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    world = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, world, port, budget, seed, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(budget + 300)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    got = sorted(q.get(timeout=5) for _ in procs)
    print("%d random programs proved by %d cooperating ranks in %.0f s: every rank's proof byte-identical to the single prover's, verify() accepted each; "
          "FRI domains %d..%d" % (got[0][1], world, budget, got[0][2][0], got[0][2][-1]))


if __name__ == "__main__":
    main()
