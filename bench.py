#!/usr/bin/env python3
"""Benchmark of the MI355X backend on BASELINE.json's metric: Goldilocks NTT field-elements/s (+ FRI.prove ms).

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

Workload (config 5 of BASELINE.json): `--total-columns` (default 8) independent trace columns of 2^24 base-field elements,
sharded over the ranks by stark_brainfuck_amd.shard (column c lives on rank c mod N: 8 / 4 / 2 / 1 columns per GPU at N = 1 / 2 /
4 / 8 -- strong scaling, the default, which is config 5 as written), resident in HBM; one step = the forward NTT of all columns of
a rank (one bfs_gl_ntt call, batch = its columns).  `--scaling weak` gives every GPU `--columns` columns instead.
value = total field elements transformed per second over all ranks, timed over exactly K steps between barrier + device
synchronisation on both sides, max over ranks.  After the timed region every rank commits to its output columns and the 64-byte
roots are all-gathered (shard.gather_roots over RCCL: the only collective of the design, never inside the timed region).
Extra keys on the same JSON line (kept under ~6 KB so that the driver's record carries every headline; per-kernel tables live in profiles/):
    roofline      dominant kernel (the NTT tile kernel) vs the 8 TB/s HBM roofline, launch time from HIP events on the kernel's stream.
                  roofline.valu / hbm_physical / bound_actual: the integer-VALU issue fraction (static PMC instruction count over THIS run's
                  step time, at the nominal clock and at the clock sampled in the sustained leg) and the bytes really moved against the tile
                  shape's copy roof; roofline.static names the tracked PMC file, the commit it was taken at and whether the kernel sources changed since
    cpu_baseline  the CPU oracle (oracle/gl_oracle.c, plain C port of ntt.py, 1 core) on a bounded sample, rank 0, N=1 only;
                  cpu_baseline.python: the same algorithm in pure Python on boxed elements at 2^14 / 2^16, timed live;
                  cpu_baseline.reference_python: the reference's own CPython figure (measured in the build container)
    sustained     the same step back to back for the seconds the CPU baseline leg takes (second thread, untimed, N = 1)
    single_column_2p24, pcie_inclusive_2p24   one 2^24-point column on its own; the same from / to pinned host memory (never `value`)
    fri_prove_ms, fri_prove_2p24_ms, merkle_tree_2p24_ms, stark_prove_ms, stark_prove_2p22_ms   headline scalars; the objects of the same
                  names (without _ms) carry the breakdowns and a nominal-clock VALU roofline each
    fri_prove.concurrent, stark_prove.concurrent   throughput mode: K = 1, 2, 4 provers on separate streams of the one GPU (threads)
    stark_prove_cooperative   N > 1: one proof carried by all ranks, timed in a separate time-limited job after the main measurement
Before the W warmup steps the device is spun up with untimed steps for --spinup-ms of wall time: after idle the
first ~10 steps run ~10 % slower while the clocks ramp, and W is chosen by the caller.
Nothing here reads /root/reference.
"""
import argparse
import ctypes
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

P = (1 << 64) - (1 << 32) + 1
SEED = 0x5EED
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def felt_array(seed, start, n):
    """felt(seed, i) = splitmix64(seed + i) mod p  (SURVEY.md 8d), vectorised."""
    with np.errstate(over="ignore"):
        x = (np.arange(start, start + n, dtype=np.uint64) + np.uint64(seed & 0xFFFFFFFFFFFFFFFF)) + np.uint64(0x9E3779B97F4A7C15)
        z = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z % np.uint64(P)


def edge_array(seed, n):
    """values next to 0, next to p and around 2^32, one in eight uniform: what trace columns look like and what drives the unreduced
    butterfly sums of the kernels over p (random operands do that once in 2^32: round 4's first lazy-sum rule passed every
    random-data test).  h = splitmix64(seed + i): kind = h & 7, small = (h >> 3) mod 6; kind 0-2: small, 3-5: p - 1 - small,
    6: h mod p, 7: 2^32 - small.  The post-run guard and tests/golden/gen_ntt24_oracle.py share this recipe."""
    with np.errstate(over="ignore"):
        x = (np.arange(0, n, dtype=np.uint64) + np.uint64(seed & 0xFFFFFFFFFFFFFFFF)) + np.uint64(0x9E3779B97F4A7C15)
        z = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        h = z ^ (z >> np.uint64(31))
    kind = h & np.uint64(7)
    small = (h >> np.uint64(3)) % np.uint64(6)
    v = np.where(kind < 3, small, np.uint64(P - 1) - small)
    v = np.where(kind == 6, h % np.uint64(P), v)
    v = np.where(kind == 7, np.uint64(1 << 32) - small, v)
    return np.ascontiguousarray(v, dtype=np.uint64)


def edge_columns(n, count=8):
    """the `count` columns of the guard's edge-value transform: edge_array with seeds 0xED6E + c, except column 1 = all p - 1 and
    column 2 = 1, p - 1, 1, p - 1, ... (sums of exactly p and 2p - 2 at the first butterfly level)"""
    cols = []
    for c in range(count):
        if c == 1:
            cols.append(np.full(n, P - 1, dtype=np.uint64))
        elif c == 2:
            cols.append(np.where(np.arange(n) % 2 == 0, np.uint64(1), np.uint64(P - 1)).astype(np.uint64))
        else:
            cols.append(edge_array(0xED6E + c, n))
    return cols


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--log-n", type=int, default=24)
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="strong: --total-columns columns sharded over the ranks (BASELINE config 5); weak: --columns per GPU")
    ap.add_argument("--total-columns", type=int, default=8, help="strong scaling: columns of the whole job")
    ap.add_argument("--columns", type=int, default=8, help="weak scaling: columns per GPU")
    ap.add_argument("--spinup-ms", type=float, default=300.0,
                    help="untimed steps run before the W warmup steps until this much wall time has passed, so that the "
                         "device clocks have ramped (the first ~10 steps after idle run ~10 %% slower); 0 disables")
    ap.add_argument("--no-fri", action="store_true")
    ap.add_argument("--no-stark", action="store_true", help="skip BrainfuckStark.prove on Hello World (config 4)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-single", action="store_true", help="skip the single-column 2^24 leg")
    ap.add_argument("--no-concurrent", action="store_true", help="skip the throughput legs (K = 1, 2, 4 provers on separate streams of this GPU)")
    ap.add_argument("--cooperative", action="store_true",
                    help="N > 1: additionally time ONE BrainfuckStark.prove carried by all ranks together (BrainfuckStark.cooperate: every rank "
                         "hashes its range of the zipped rows; opt-in, the default legs run independent replicas)")
    ap.add_argument("--no-cooperative", action="store_true",
                    help="N > 1: do not time the cooperative proof at all (by default rank 0 runs it as a SEPARATE, time-limited job after "
                         "the main measurement -- see guarded_cooperative -- unless --cooperative already ran it in-job)")
    ap.add_argument("--coop-leg", action="store_true", help=argparse.SUPPRESS)     # internal: the separate job of guarded_cooperative
    ap.add_argument("--stark-worker", type=float, default=0.0, help=argparse.SUPPRESS)   # internal: one prover PROCESS of bench_stark_concurrent
    ap.add_argument("--no-tune", action="store_true", help="do not call bfs_ntt_tune on the step's buffer pair (every step takes the direct route)")
    ap.add_argument("--no-check", action="store_true", help="skip the post-run round-trip check and root gather (PMC collection runs)")
    args = ap.parse_args()

    if args.stark_worker > 0:
        raise SystemExit(stark_worker(args.stark_worker))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves -- the same one-process-per-GPU launch the driver uses
        # (torch.distributed.run over 127.0.0.1); rank 0 of that job prints the JSON line on the stdout we share with it
        raise SystemExit(self_launch(args.gpus))
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d): launch with torch.distributed.run --nproc-per-node N" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback for the hot path")
    # test hooks (tests/test_gpu_distributed.py runs two ranks on the one GPU of a gpurun box): BFS_BENCH_BACKEND=gloo exchanges CPU
    # tensors instead of going through RCCL, BFS_BENCH_DEVICE pins every rank to that device.  The driver sets neither.
    backend = os.environ.get("BFS_BENCH_BACKEND", "nccl")
    if "BFS_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["BFS_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    coll_device = torch.device("cuda", local_rank) if backend == "nccl" else torch.device("cpu")
    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ or os.environ.get("BFS_BENCH_FORCE_DIST"):
        # launched by torch.distributed.run: bring up RCCL even for a single rank so that the collective path is the one exercised
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # this RCCL build prints a version banner on stdout when the first communicator comes up; stdout carries exactly one JSON
        # line, so the banner is sent to stderr (file-descriptor level: it is written by native code)
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        # A communicator that does not come up must FAIL the job, loudly and soon, not hang it: a rank stuck inside RCCL's bootstrap cannot
        # be interrupted from Python, so a watchdog thread ends the process (exit status 3, a line on stderr saying which rank waited for
        # what) when rendezvous + the first collective have not finished within BFS_BENCH_COMM_TIMEOUT_S (120 s; they take < 5 s on a
        # healthy node).  torch.distributed.run then tears the other ranks down and the driver sees a non-zero exit.
        import datetime
        import threading
        limit = float(os.environ.get("BFS_BENCH_COMM_TIMEOUT_S", "120"))
        stage = ["rendezvous (init_process_group)"]

        def give_up():
            sys.stderr.write("bench.py: rank %d of %d (device %d): the %s communicator did not come up within %.0f s, stuck in %s -- aborting\n"
                             % (rank, world, local_rank, backend, limit, stage[0]))
            sys.stderr.flush()
            os._exit(3)
        watchdog = threading.Timer(limit, give_up)
        watchdog.daemon = True
        watchdog.start()
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank),
                                        timeout=datetime.timedelta(seconds=limit))
            else:
                dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=limit))
            stage[0] = "the first collective (barrier: RCCL builds its rings here)"
            dist.barrier()             # RCCL builds its communicators on the first collective (100s of ms): pay that here, not
            torch.cuda.synchronize()   # between the clock spin-up and the timed region, where the idle GPU would clock down again
            watchdog.cancel()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    from stark_brainfuck_amd import _lib, shard
    from stark_brainfuck_amd.device import DeviceBuffer, DeviceView
    lib = _lib.load()
    _lib.check(lib.bfs_set_device(local_rank))
    if args.coop_leg:
        # the separate job of guarded_cooperative: nothing but the cooperative proof; rank 0 prints its own one-line JSON
        coop = bench_stark_cooperative(world, rank, coll_device if backend == "nccl" else None, dist, torch) if dist is not None else None
        if rank == 0:
            print(json.dumps({"stark_prove_cooperative": coop}), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    # who is in the job, as the collective backend itself sees it: world size from the process group and every rank's device
    # (index + PCI bus id), all-gathered -- so the line shows N distinct GPUs behind N ranks, not N ranks on one device
    ranks_seen, rank_devices = 1, [device_identity(torch, local_rank)]
    if dist is not None:
        ranks_seen = dist.get_world_size()
        gathered = [None] * world
        dist.all_gather_object(gathered, rank_devices[0])
        rank_devices = gathered

    log_n = args.log_n
    n = 1 << log_n
    root = lib.bfs_gl_primitive_root(log_n)
    total_cols = args.total_columns if args.scaling == "strong" else args.columns * world
    my_cols = shard.assign_columns(total_cols, world, rank)          # column c -> rank c mod world (the product's sharding)
    cols = len(my_cols)
    if cols == 0:
        raise SystemExit("rank %d owns no column: --total-columns (%d) must be >= the number of GPUs (%d)" % (rank, total_cols, world))
    host_in = np.concatenate([felt_array(SEED + (c << 32), 0, n) for c in my_cols])
    d_in = DeviceBuffer.from_numpy(host_in)
    d_out = DeviceBuffer(n * cols)
    stream = 0

    def step():
        _lib.check(lib.bfs_gl_ntt(d_in.ptr, n, n, d_out.ptr, n, log_n, cols, root, 1, 1, stream))

    # The step repeats one (input, output) pair of buffers: the explicit, documented tuning call of the library (bfs_ntt_tune: ~63 ms,
    # untimed, before everything) chooses where the first pass writes.  bfs_gl_ntt itself never measures anything (round 5); the same
    # K steps on the DIRECT route are timed after the headline region and printed beside it (ms_per_step_direct_route).
    tuned_route = None
    if not args.no_tune:
        r = ctypes.c_int(-1)
        _lib.check(lib.bfs_ntt_tune(d_in.ptr, n, d_out.ptr, n, log_n, cols, root, stream, ctypes.byref(r)))
        tuned_route = r.value

    def sync_all():
        _lib.check(lib.bfs_stream_synchronize(stream))
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # the CPU leg first (rank 0, N = 1).  While the host core works through it, the GPU runs the same NTT step back to back from a
    # second thread: an untimed SUSTAINED leg (seconds instead of the milliseconds of the timed region: clocks and temperature at
    # their steady state, and the GPU visibly busy to whoever samples rocm-smi during the run); the timed region follows it.
    cpu_line, sustained = None, None
    if rank == 0 and world == 1 and not args.no_cpu:
        import threading
        stop, count = threading.Event(), [0, 0.0]

        def sustain():
            t0 = time.perf_counter()
            while not stop.is_set():
                for _ in range(32):
                    step()
                _lib.check(lib.bfs_stream_synchronize(stream))
                count[0] += 32
            count[1] = time.perf_counter() - t0
        worker = threading.Thread(target=sustain)
        worker.start()
        clock = {}
        sampler = threading.Thread(target=lambda: clock.update(sample_clock_under_load(local_rank, delay_s=3.0) or {}))
        sampler.start()
        try:
            cpu_line = cpu_baseline(log_n)
        finally:
            sampler.join()
            stop.set()
            worker.join()
        cpu_line["python"] = cpu_baseline_python()
        sustained = {"seconds": count[1], "steps": count[0], "ms_per_step": count[1] / max(count[0], 1) * 1e3,
                     "elements_per_s": n * cols * count[0] / max(count[1], 1e-9),
                     "note": "untimed: the NTT step back to back from a second thread for as long as the CPU baseline leg runs"}
        if clock:
            sustained.update(clock)
    spin_t0, spin_steps = time.perf_counter(), 0
    while args.spinup_ms > 0 and (time.perf_counter() - spin_t0) * 1e3 < args.spinup_ms:
        step()
        _lib.check(lib.bfs_stream_synchronize(stream))
        spin_steps += 1
    for _ in range(args.warmup):
        step()
    ev0, ev1 = ctypes.c_void_p(), ctypes.c_void_p()
    lib.bfs_event_create(ctypes.byref(ev0)); lib.bfs_event_create(ctypes.byref(ev1))
    sync_all()
    t0 = time.perf_counter()
    lib.bfs_event_record(ev0, stream)
    for _ in range(args.steps):
        step()
    lib.bfs_event_record(ev1, stream)
    _lib.check(lib.bfs_stream_synchronize(stream))
    torch.cuda.synchronize()
    t1 = time.perf_counter()        # this rank's K steps are done; the closing barrier below is not part of its work
    sync_all()
    elapsed = t1 - t0               # MAX over ranks is taken below
    kern_ms = ctypes.c_float()
    _lib.check(lib.bfs_event_elapsed_ms(ev0, ev1, ctypes.byref(kern_ms)))
    if dist is not None:
        t = torch.tensor([elapsed, kern_ms.value], dtype=torch.float64, device=coll_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kern = float(t[0]), float(t[1])
    else:
        kern = kern_ms.value

    # ---- the same K steps on the direct route (what a caller who never calls bfs_ntt_tune gets): remembered pairs forgotten, W warm-up
    # steps, K timed ones between HIP events
    direct_ms = None
    if tuned_route is not None:
        _lib.check(lib.bfs_ntt_route_forget(None, None))
        for _ in range(max(args.warmup, 1)):
            step()
        lib.bfs_event_record(ev0, stream)
        for _ in range(args.steps):
            step()
        lib.bfs_event_record(ev1, stream)
        _lib.check(lib.bfs_stream_synchronize(stream))
        dm = ctypes.c_float()
        _lib.check(lib.bfs_event_elapsed_ms(ev0, ev1, ctypes.byref(dm)))
        direct_ms = dm.value / args.steps
        if dist is not None:
            t = torch.tensor([direct_ms], dtype=torch.float64, device=coll_device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            direct_ms = float(t[0])

    # ---- after the timed region: correctness guard + the one collective of the design (roots of all columns)
    from stark_brainfuck_amd.arrays import BaseArray
    from stark_brainfuck_amd.merkle import Merkle
    local_roots, column_sha, guard = {}, {}, None
    if not args.no_check:
        # every column of this rank, in full: (1) inverse transform of the whole batch == the input, (2) one SHA-256 per output
        # column, compared with the oracle's known answers when the workload is the fixture's (2^24, tests/golden/ntt24_oracle.json:
        # same seed, same columns), (3) a Merkle tree over the whole column (2^log_n leaves), whose root is what gets all-gathered
        import hashlib
        inv = DeviceBuffer(n * cols)
        _lib.check(lib.bfs_gl_ntt(d_out.ptr, n, n, inv.ptr, n, log_n, cols, lib.bfs_gl_inv(root), 1, lib.bfs_gl_inv(n), stream))
        back = inv.to_numpy()
        assert (back == host_in).all(), "intt(ntt(x)) != x on the bench data"
        del back, inv
        known = None
        gpath = os.path.join(ROOT, "tests", "golden", "ntt24_oracle.json")
        if log_n == 24 and os.path.exists(gpath):
            g = json.load(open(gpath))
            if g.get("seed") == SEED and g.get("root") == root:
                known = g["columns"]
        checked = 0
        for j, c in enumerate(my_cols):
            col = d_out.to_numpy(n, offset=j * n)
            column_sha[c] = hashlib.sha256(np.ascontiguousarray(col, dtype="<u8").tobytes()).hexdigest()
            if known is not None and c < len(known):
                assert column_sha[c] == known[c]["output_sha256"], "column %d differs from the oracle's known answer" % c
                checked += 1
            tree = Merkle(BaseArray(DeviceView(d_out, j * n, n), n))       # leaves hashed where the transform left them
            local_roots[c] = tree.root()
            del tree, col
        # (4) rank 0: the same call shape (8 columns of 2^24, same buffers, same route) on EDGE values -- operands next to 0, p and 2^32,
        # which random columns never contain and which the kernels' unreduced butterfly sums must survive -- against the oracle's known
        # answers for them (tests/golden/ntt24_oracle.json: edge_columns, made by gen_ntt24_oracle.py with edge_columns() above)
        edge_checked = 0
        if rank == 0 and known is not None and g.get("edge_columns") and cols == len(g["edge_columns"]):
            edge_in = np.concatenate(edge_columns(n, cols))
            _lib.check(lib.bfs_memcpy_h2d(d_in.ptr, edge_in.ctypes.data, 8 * n * cols, stream))
            del edge_in
            step()
            for j in range(cols):
                col = d_out.to_numpy(n, offset=j * n)
                sha = hashlib.sha256(np.ascontiguousarray(col, dtype="<u8").tobytes()).hexdigest()
                assert sha == g["edge_columns"][j]["output_sha256"], "edge-value column %d differs from the oracle's known answer" % j
                edge_checked += 1
                del col
            _lib.check(lib.bfs_memcpy_h2d(d_in.ptr, host_in.ctypes.data, 8 * n * cols, stream))
        guard = {"columns_round_tripped": cols, "columns_sha256_vs_oracle_known_answers": checked,
                 "edge_value_columns_vs_oracle_known_answers": edge_checked, "merkle_leaves_per_column": n}
    world_roots = None
    if not args.no_check:
        # the one collective of the design: all-gather of the per-column roots (RCCL over xGMI when world > 1)
        world_roots = shard.gather_roots(local_roots, total_cols, world, rank, device=coll_device if (dist is not None and backend == "nccl") else None)
        assert len(world_roots) == total_cols and all(len(r) == 64 for r in world_roots)
        assert all(world_roots[c] == local_roots[c] for c in my_cols)

    elems = n * total_cols * args.steps
    value = elems / elapsed
    line = {
        "metric": "goldilocks_ntt_field_elements_per_sec",
        "value": value,
        "unit": "elements/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "u64 (mod 2^64-2^32+1)",
        "data": "synthetic (splitmix64 mod p, seed 0x5EED)",
        "config": {"workload": "forward NTT of %d independent 2^%d-point base-field columns resident in HBM (BASELINE config 5)" % (total_cols, log_n),
                   "log_n": log_n, "total_columns": total_cols, "columns_per_gpu": shard.columns_per_rank(total_cols, world),
                   "parallelism": "column c on rank c mod %d, no data-path collective; roots all-gathered after the timed region" % world},
        "roots_sha256": __import__("hashlib").sha256(b"".join(world_roots)).hexdigest() if world_roots else None,
        "guard": guard,
        "rccl_ranks_seen": ranks_seen, "collective_backend": (backend if dist is not None else None), "rank_devices": rank_devices,
        "algorithmic_GBps": 16.0 * elems / elapsed / 1e9,
        "clock_spinup": {"ms": args.spinup_ms, "untimed_steps": spin_steps},
        "ms_per_step_direct_route": direct_ms,
        "ntt_tune": None if tuned_route is None else ("direct" if tuned_route < 0 else "buffer%d" % tuned_route),
    }
    # FRI replicas: one independent Fri.prove per GPU at the same time (a single FRI instance is sequential in its rounds
    # and is not sharded); every rank reports, rank 0 prints the slowest and the aggregate rate
    fri_all = None
    if not args.no_fri:
        mine = bench_fri(lib, _lib, stream, 18)
        fri_all = [mine["ms"]]
        if dist is not None:
            t = torch.tensor([mine["ms"]], dtype=torch.float64, device=coll_device)
            parts = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(parts, t)
            fri_all = [float(p[0]) for p in parts]
    # STARK prover replicas, the same way: every GPU proves the Hello-World program on its own
    stark_mine, stark_all = None, None
    if not args.no_stark and not args.no_fri:
        stark_mine = bench_stark()
        stark_all = [stark_mine["ms"]]
        if dist is not None:
            t = torch.tensor([stark_mine["ms"]], dtype=torch.float64, device=coll_device)
            parts = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(parts, t)
            stark_all = [float(p[0]) for p in parts]
    coop = None
    if args.cooperative and dist is not None and world > 1 and (world & (world - 1)) == 0:
        coop = bench_stark_cooperative(world, rank, coll_device if backend == "nccl" else None, dist, torch)
    if rank == 0:
        # dominant kernel = the NTT tile kernel (npass launches per step); algorithmic bytes of one launch =
        # 16 B/element * n * columns / npass (DESIGN.md "Roofline accounting"); HIP events bracket exactly K steps
        npass = 1 if log_n <= 12 else (log_n + 7) // 8
        launches = npass * args.steps
        avg_launch_s = kern * 1e-3 / launches
        step_s = kern * 1e-3 / args.steps
        bytes_per_launch = 16.0 * n * cols / npass
        achieved = bytes_per_launch / avg_launch_s / 1e9
        # HBM traffic and VALU instruction counts are NOT measured by this run (bench.py runs no profiler): they are the tracked PMC
        # summary of the NTT-only command (tools/prof_ntt.sh -> profiles/ntt_traffic.json: separate rocprofv3 --pmc passes, gfx950
        # FETCH_SIZE correction), labelled static with the commit they were taken at; the times they are divided by are this run's
        traffic, static, valu, hbm_phys = None, None, None, None
        tpath = os.path.join(ROOT, "profiles", "ntt_traffic.json")
        if os.path.exists(tpath):
            t = json.load(open(tpath))
            if t.get("log_n") == log_n and t.get("columns") == cols:
                traffic = t["hbm_bytes_per_launch"]
                static = {"file": "profiles/ntt_traffic.json", "taken_at": t.get("source_commit"), "kernel_sources_sha16": t.get("kernel_sources_sha16"),
                          "current_sources_sha16": kernel_sources_sha16()}          # (static: roofline.traffic and valu.wave_instructions_per_step)
                static["stale"] = bool(static["kernel_sources_sha16"]) and static["kernel_sources_sha16"] != static["current_sources_sha16"]
                if not static["stale"]:
                    del static["current_sources_sha16"]          # (equal: said once)
                instr = t["valu_wave_instructions_per_step"]
                sclk = (sustained or {}).get("sclk_mhz_under_load")
                valu = {"wave_instructions_per_step": instr, "instructions_per_element": instr * 64.0 / (n * cols),
                        "issue_frac_at_nominal_2400mhz": instr * 4.0 / SIMDS / (NOMINAL_SCLK_MHZ * 1e6) / step_s}
                if sclk:
                    # the package sits at its power limit under this step: the VALU roof is 1024 SIMDs x THIS clock / 4 cycles per wave instruction
                    valu["sclk_mhz_sampled_in_sustained_leg"] = sclk
                    f = instr * 4.0 / SIMDS / (sclk * 1e6) / step_s
                    if f <= 1.0:
                        valu["issue_frac_at_sampled_sclk"] = f
                    else:       # one rocm-smi sample of a clock that differs between XCDs and moments: not evidence when it prices the step above 1
                        valu["sclk_sample_inconsistent_with_step_time"] = True
                rate = traffic / avg_launch_s / 1e9
                COPY_ROOF_GBS = 5200.0      # a kernel that only MOVES this tile shape (256 rows x 128 B, non-temporal): profiles/r02/microbench_tile_copy_roof.txt
                hbm_phys = {"GBps": rate, "frac_of_peak": rate / HBM_PEAK_GBS, "frac_of_tile_copy_roof_5200": rate / COPY_ROOF_GBS}
        bound_actual = None
        if valu and hbm_phys:
            vf = valu.get("issue_frac_at_sampled_sclk", valu["issue_frac_at_nominal_2400mhz"])
            bound_actual = "this run: VALU issue %.2f of its roof (%s clock), physical HBM %.2f of the tile shape's copy roof" % (
                vf, "sampled" if "issue_frac_at_sampled_sclk" in valu else "nominal", hbm_phys["frac_of_tile_copy_roof_5200"])
        line["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                            "traffic": traffic, "wasted_traffic_ratio": (traffic / bytes_per_launch) if traffic else None,
                            "kernel": "ntt_tile_kernel_split<4,4,0,4,MODE,NT>",
                            "launches_per_step": npass, "avg_launch_ms": avg_launch_s * 1e3, "algorithmic_bytes_per_launch": bytes_per_launch,
                            "valu": valu, "hbm_physical": hbm_phys, "static": static,
                            "route_probe": route_probe_info(lib)}
        if log_n == 24 and not args.no_single:
            line["single_column_2p24"] = bench_single_column(lib, _lib, d_in, d_out, n, log_n, root, stream)
            line["single_column_2p24_ms"] = line["single_column_2p24"]["ms"]
            line["pcie_inclusive_2p24"] = bench_pcie_inclusive(lib, _lib, d_in, d_out, n, log_n, root, stream)
            # the other configurations BASELINE.json names, as scalars beside the headline (round-5 verdict, next #4)
            line["other_shapes"] = bench_other_shapes(lib, _lib, d_in, d_out, n, cols, root, stream)
            for k in ("ntt_2p20_fwd_inv_us", "lde_2p24_x4_ms", "intt_8x2p24_ms"):
                line[k] = line["other_shapes"][k].pop("value")          # the scalars sit at the top level; `other_shapes` keeps the fractions
        if not args.no_fri:
            line["fri_prove_ms"] = max(fri_all)
            line["fri_prove"] = mine
            line["fri_prove"]["replicas"] = {"per_gpu_ms": [round(v, 4) for v in fri_all], "proofs_per_s": world / (max(fri_all) * 1e-3)}
            line["fri_prove"]["roofline"] = fri_roofline(line["fri_prove"])
            line["fri_prove_2p24"] = bench_fri(lib, _lib, stream, 22)
            line["fri_prove_2p24_ms"] = line["fri_prove_2p24"]["ms"]
            line["fri_prove_2p24"]["roofline"] = fri_roofline(line["fri_prove_2p24"])
            line["merkle_tree_2p24"] = bench_merkle_tree(lib, _lib, stream, 24)
            line["merkle_tree_2p24_ms"] = line["merkle_tree_2p24"]["ms"]
            if world == 1 and not args.no_concurrent:
                line["fri_prove"]["concurrent"] = bench_fri_concurrent(lib, _lib, 18)
        if not args.no_stark and not args.no_fri:
            line["stark_prove_ms"] = max(stark_all)
            line["stark_prove"] = stark_mine
            line["stark_prove"]["replicas"] = {"per_gpu_ms": [round(v, 4) for v in stark_all], "proofs_per_s": world / (max(stark_all) * 1e-3)}
            line["stark_prove_2p22"] = bench_stark("+" * 64 + "[>" + "+" * 64 + "[>++++<-]<-]+++.", "nested loops, 37 254 cycles")
            line["stark_prove_2p22_ms"] = line["stark_prove_2p22"]["ms"]
            line["stark_prove_2p22"]["roofline"] = stark_roofline(line["stark_prove_2p22"])
            if world == 1 and not args.no_concurrent:
                line["stark_prove"]["concurrent"] = bench_stark_concurrent()
            line["per_kernel_tables"] = "profiles/prover_valu.json (static; tools/prof_prover.sh)"
        if coop is not None:
            line["stark_prove_cooperative"] = coop
        if sustained is not None:
            line["sustained"] = sustained
        if cpu_line is not None:
            line["cpu_baseline"] = cpu_line
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if coop is None and world > 1 and (world & (world - 1)) == 0 and not args.no_cooperative and not args.no_stark and not args.no_fri:
            del d_in, d_out
            line["stark_prove_cooperative"] = guarded_cooperative(world)
        print(json.dumps(slim(line, world), separators=(",", ":")), flush=True)


def route_probe_info(lib):
    """what bfs_ntt_tune read in this process (ntt.hip: ntt_measure_route; called once on the step's buffer pair before anything is
    timed): passes 0 + 1 straight into the output and through each of three library buffers, and which it kept --
    when all four read alike the box has no fast pair to offer and the step is what it is (profiles/r04/ab_ws_probe.txt)"""
    import ctypes
    us, route, probes = (ctypes.c_float * 4)(), ctypes.c_int(-1), ctypes.c_ulonglong(0)
    if lib.bfs_ntt_route_probe_info(us, ctypes.byref(route), ctypes.byref(probes)) != 0 or probes.value == 0:
        return {"probes": 0}
    return {"probes": int(probes.value), "passes_0_1_us": [round(us[0], 1), round(us[1], 1), round(us[2], 1), round(us[3], 1)],       # direct, buffers 0-2
            "chosen": "direct" if route.value < 0 else "buffer%d" % route.value}


def slim(line, world):
    """the printed form of the line: explanatory strings (every `note` / `model`: the module docstring says what each leg is) and
    single-GPU replica lists dropped, floats rounded to 6 significant digits -- the driver keeps an 8 KB tail of stdout and every
    headline has to be inside it (round-3 verdict #4)"""
    def walk(v, key=None):
        if isinstance(v, dict):
            out = {}
            for k, x in v.items():
                if k in ("note", "model") or (k == "replicas" and world == 1) or (k == "rank_devices" and world == 1):
                    continue
                out[k] = walk(x, k)
            return out
        if isinstance(v, list):
            return [walk(x) for x in v]
        if isinstance(v, float):
            return float("%.6g" % v)
        return v
    return walk(line)


def device_identity(torch, index):
    try:
        props = torch.cuda.get_device_properties(index)
        bus = getattr(props, "pci_bus_id", None)
        return {"device": index, "name": props.name, "pci_bus_id": bus, "uuid": str(getattr(props, "uuid", "")) or None}
    except Exception:
        return {"device": index}


def guarded_cooperative(n_ranks, limit_s=300):
    """ONE proof carried by all GPUs together (BrainfuckStark.cooperate), timed in a job of its own: `bench.py --gpus N --coop-leg` under
    torch.distributed.run, started by rank 0 after the main measurement is complete and its process group is gone, with a time limit.
    It is the one leg whose collectives (row-range commitments, all-gather of the combination codeword over the library's device
    memory) have never run across real xGMI links in development -- a failure or hang there must cost this entry, not the line."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK",
                                                              "MASTER_ADDR", "MASTER_PORT") and not k.startswith("TORCHELASTIC")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), "--gpus", str(n_ranks), "--coop-leg"]
    t0 = time.perf_counter()
    import signal
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out_text, err_text = proc.communicate(timeout=limit_s)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(proc.pid, signal.SIGKILL)          # the launcher AND its ranks (its own session: nothing of ours is in that group)
        except OSError:
            pass
        proc.communicate()
        return {"error": "no result within %d s (job killed)" % limit_s, "ranks": n_ranks, "separate_job": True}

    class res:
        stdout, stderr, returncode = out_text, err_text, proc.returncode
    for l in reversed(res.stdout.splitlines()):
        if l.startswith("{"):
            try:
                out = json.loads(l).get("stark_prove_cooperative")
            except ValueError:
                continue
            if out:
                out["separate_job"] = True
                out["job_seconds"] = round(time.perf_counter() - t0, 1)
                return out
    return {"error": "exit status %d: %s" % (res.returncode, (res.stderr or res.stdout)[-400:]), "ranks": n_ranks, "separate_job": True}


def self_launch(n_ranks):
    """re-executes this command line under torch.distributed.run with one rank per GPU and returns its exit status"""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")         # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    return subprocess.call(cmd, env=env)


def bench_single_column(lib, _lib, d_in, d_out, n, log_n, root, stream, steps=200):
    """one 2^24-point column on its own (128 MiB in, 128 MiB out: the transform north_star's target sentence is about)"""
    def one():
        _lib.check(lib.bfs_gl_ntt(d_in.ptr, n, n, d_out.ptr, n, log_n, 1, root, 1, 1, stream))
    for _ in range(20):
        one()
    e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
    lib.bfs_event_create(ctypes.byref(e0)); lib.bfs_event_create(ctypes.byref(e1))
    lib.bfs_event_record(e0, stream)
    for _ in range(steps):
        one()
    lib.bfs_event_record(e1, stream)
    _lib.check(lib.bfs_stream_synchronize(stream))
    ms = ctypes.c_float()
    _lib.check(lib.bfs_event_elapsed_ms(e0, e1, ctypes.byref(ms)))
    per = ms.value / steps
    gbs = 16.0 * n / per / 1e6
    return {"ms": per, "elements_per_s": n / per * 1e3, "algorithmic_GBps": gbs, "frac_of_hbm_peak": gbs / HBM_PEAK_GBS, "steps": steps,
            "note": "forward NTT of one column, batch 1, HIP events over %d back-to-back transforms" % steps}


def _timed(lib, _lib, stream, call, steps, warm=10):
    for _ in range(warm):
        call()
    e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
    lib.bfs_event_create(ctypes.byref(e0)); lib.bfs_event_create(ctypes.byref(e1))
    lib.bfs_event_record(e0, stream)
    for _ in range(steps):
        call()
    lib.bfs_event_record(e1, stream)
    _lib.check(lib.bfs_stream_synchronize(stream))
    ms = ctypes.c_float()
    _lib.check(lib.bfs_event_elapsed_ms(e0, e1, ctypes.byref(ms)))
    lib.bfs_event_destroy(e0); lib.bfs_event_destroy(e1)
    return ms.value / steps


def bench_other_shapes(lib, _lib, d_in, d_out, n, cols, root, stream):
    """BASELINE.json's other NTT shapes on the buffers of the headline step, HIP events, `frac` on 16 bytes per OUTPUT element:
       ntt_2p20_fwd_inv_us   config 2: one 2^20 column forward, then inverse (ntt.py:4-42; the reference: 242.7 s + 236.2 s)
       lde_2p24_x4_ms        the shape Table.lde makes (table.py:138-149 -> fri.py:26-30): four columns of 2^22 coefficients, zero-padded and
                             evaluated on the coset of 2^24 points (ntt.py:164-168) in one call
       intt_8x2p24_ms        the inverse of the headline step: eight 2^24 columns, omega^-1, n^-1 folded in (ntt.py:26-42)"""
    out = {}
    n20 = 1 << 20
    w20, w20i, n20i = lib.bfs_gl_primitive_root(20), lib.bfs_gl_inv(lib.bfs_gl_primitive_root(20)), lib.bfs_gl_inv(n20)

    def pair20():
        _lib.check(lib.bfs_gl_ntt(d_in.ptr, n20, n20, d_out.ptr, n20, 20, 1, w20, 1, 1, stream))
        _lib.check(lib.bfs_gl_ntt(d_out.ptr, n20, n20, d_out.ptr + 8 * n20, n20, 20, 1, w20i, 1, n20i, stream))
    ms = _timed(lib, _lib, stream, pair20, 200)
    out["ntt_2p20_fwd_inv_us"] = {"value": ms * 1e3, "frac": 2 * 16.0 * n20 / ms / 1e6 / HBM_PEAK_GBS, "reference_s": [242.7, 236.2]}
    quarter = n // 4
    lde_cols = min(4, cols)

    def lde():
        _lib.check(lib.bfs_gl_ntt(d_in.ptr, quarter, n, d_out.ptr, n, 24, lde_cols, root, 7, 1, stream))
    ms = _timed(lib, _lib, stream, lde, 30)
    out["lde_2p24_x4_ms"] = {"value": ms, "frac": 16.0 * n * lde_cols / ms / 1e6 / HBM_PEAK_GBS}
    wi, ni = lib.bfs_gl_inv(root), lib.bfs_gl_inv(n)

    def inverse():
        _lib.check(lib.bfs_gl_ntt(d_in.ptr, n, n, d_out.ptr, n, 24, cols, wi, 1, ni, stream))
    ms = _timed(lib, _lib, stream, inverse, 20)
    out["intt_8x2p24_ms"] = {"value": ms, "frac": 16.0 * n * cols / ms / 1e6 / HBM_PEAK_GBS}
    return out


def bench_pcie_inclusive(lib, _lib, d_in, d_out, n, log_n, root, stream, reps=5):
    """the same transform when the boundary hands over HOST buffers (never `value`, which is quoted with inputs resident in HBM): one
    2^log_n column from pinned host memory to the GPU, transformed, and back into pinned host memory -- 16 bytes per element over PCIe
    around 16 algorithmic bytes per element in HBM."""
    from stark_brainfuck_amd.device import pinned_empty
    h_in, h_out = pinned_empty(n), pinned_empty(n)
    h_in[:] = felt_array(SEED + 5, 0, n)
    times = []
    for _ in range(reps):
        t0 = time.perf_counter()
        _lib.check(lib.bfs_memcpy_h2d(d_in.ptr, h_in.ctypes.data, 8 * n, stream))
        _lib.check(lib.bfs_gl_ntt(d_in.ptr, n, n, d_out.ptr, n, log_n, 1, root, 1, 1, stream))
        _lib.check(lib.bfs_memcpy_d2h(h_out.ctypes.data, d_out.ptr, 8 * n, stream))
        _lib.check(lib.bfs_stream_synchronize(stream))
        times.append(time.perf_counter() - t0)
    t = statistics.median(times[1:])
    return {"ms": t * 1e3, "elements_per_s": n / t, "pcie_bytes": 16 * n, "pcie_GBps_if_the_transform_were_free": 16.0 * n / t / 1e9, "columns": 1,
            "note": "pinned host -> HBM, forward NTT, HBM -> pinned host, one 2^%d column per call, serial" % log_n}


def kernel_sources_sha16():
    """what the static PMC figures (profiles/ntt_traffic.json) must have been taken on: the NTT kernels' sources"""
    import hashlib
    h = hashlib.sha256()
    for f in ("gl.hpp", "ntt_core.hpp", "ntt_plan.hpp", "ntt.hip"):
        h.update(open(os.path.join(ROOT, "stark_brainfuck_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def _throughput(workers, seconds):
    """runs `workers` (callables returning after ONE proof) in one thread each, back to back for ~`seconds`; returns proofs per second
    over the wall clock from the common start to the last thread's end"""
    import threading
    start = threading.Barrier(len(workers) + 1)
    counts, errors = [0] * len(workers), []

    def loop(i):
        try:
            workers[i]()                      # one untimed proof: streams, scratch areas and caches of this thread exist afterwards
            start.wait()
            t_end = time.perf_counter() + seconds
            while time.perf_counter() < t_end:
                workers[i]()
                counts[i] += 1
        except Exception as e:                # noqa: BLE001
            errors.append(repr(e))
            try:
                start.abort()
            except Exception:                 # noqa: BLE001
                pass
    threads = [threading.Thread(target=loop, args=(i,)) for i in range(len(workers))]
    for th in threads:
        th.start()
    try:
        start.wait()
    except threading.BrokenBarrierError:
        pass
    t0 = time.perf_counter()
    for th in threads:
        th.join()
    wall = time.perf_counter() - t0
    if errors:
        return {"error": errors[0][:200]}
    return {"proofs": sum(counts), "seconds": round(wall, 3), "proofs_per_s": sum(counts) / wall}


def bench_fri_concurrent(lib, _lib, log_d, ks=(1, 2, 4), seconds=0.6):
    """throughput mode (round-3 verdict #3): K provers at once on ONE GPU, each a thread with its own bfs_stream_create stream and its own
    copy of the codeword, Fri.prove (N = 4 * 2^log_d) back to back.  A single proof is a chain of dependent launches and host round trips
    that leaves most of the chip idle; this is what concurrency recovers.  Every prover's transcript must be the K = 1 transcript."""
    import hashlib
    from stark_brainfuck_amd.device import DeviceBuffer
    expansion, t = 4, 4
    d, N = 1 << log_d, (1 << log_d) * expansion
    log_N = N.bit_length() - 1
    omega = lib.bfs_gl_primitive_root(log_N)
    coeffs = felt_array(SEED, 0, 3 * d).reshape(d, 3).T.copy()
    d_coef = DeviceBuffer.from_numpy(coeffs.reshape(-1))
    out = {}
    digests = set()
    for K in ks:
        streams, cws = [], []
        for _ in range(K):
            st = ctypes.c_void_p()
            _lib.check(lib.bfs_stream_create(ctypes.byref(st)))
            cw = DeviceBuffer(3 * N)
            _lib.check(lib.bfs_gl_ntt(d_coef.ptr, d, d, cw.ptr, N, log_N, 3, omega, 7, 1, st))
            _lib.check(lib.bfs_stream_synchronize(st))
            streams.append(st); cws.append(cw)
        last = [None] * K

        def make(i):
            def one():
                ps = lib.bfs_ps_new()
                idx = (ctypes.c_uint64 * t)()
                _lib.check(lib.bfs_fri_prove(ps, cws[i].ptr, N, log_N, 7, omega, expansion, t, idx, streams[i]))
                _lib.check(lib.bfs_stream_synchronize(streams[i]))
                if last[i] is None:
                    size = ctypes.c_size_t()
                    _lib.check(lib.bfs_ps_serialize(ps, 1 << 62, None, 0, ctypes.byref(size)))
                    buf = ctypes.create_string_buffer(size.value)
                    _lib.check(lib.bfs_ps_serialize(ps, 1 << 62, buf, size.value, ctypes.byref(size)))
                    last[i] = hashlib.sha256(buf.raw[:size.value]).hexdigest()
                lib.bfs_ps_free(ps)
            return one
        r = _throughput([make(i) for i in range(K)], seconds)
        digests.update(last)
        out[str(K)] = round(r["proofs_per_s"], 1) if "proofs_per_s" in r else r
        for st in streams:
            lib.bfs_stream_destroy(st)
        for cw in cws:
            cw.free()
    base = out.get("1")
    return {"proofs_per_s_by_provers": out, "best_over_single": (max(v for v in out.values() if isinstance(v, float)) / base) if isinstance(base, float) else None,
            "transcripts_identical_across_provers": len(digests) == 1, "N": N,
            "note": "K threads x (own stream, own codeword), bfs_fri_prove back to back for %.1f s per K" % seconds}


def stark_worker(seconds):
    """one prover process of bench_stark_concurrent's process mode: proves Hello World once, reports `ready`, waits for `go` on stdin,
    then proves back to back for `seconds` and prints how many proofs it wrote and the SHA-256 of the first"""
    import hashlib
    from stark_brainfuck_amd import randomness
    from stark_brainfuck_amd.brainfuck_stark import BrainfuckStark
    from stark_brainfuck_amd.vm import VirtualMachine
    program = VirtualMachine.compile(HELLO_WORLD)
    running_time, inputs, outputs = VirtualMachine.run(program)
    mats = VirtualMachine.simulate(program, input_data=inputs)

    def one():
        with randomness.override(FixedRandomness()):
            return BrainfuckStark(running_time, len(mats[1]), program, inputs, outputs).prove(program, *mats)
    first = hashlib.sha256(one()).hexdigest()
    one()
    print("ready", flush=True)
    if sys.stdin.readline().strip() != "go":
        return 1
    count, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        one()
        count += 1
    print(json.dumps({"proofs": count, "seconds": time.perf_counter() - t0, "sha256": first}), flush=True)
    return 0


def stark_process_throughput(K, seconds):
    """K prover PROCESSES on this GPU (no shared interpreter lock): started, warmed up, released together; proofs per second over
    the slowest process' window"""
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--stark-worker", str(seconds)], stdin=subprocess.PIPE,
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env) for _ in range(K)]
    import threading
    watchdog = threading.Timer(90.0, lambda: [p.kill() for p in procs])      # a stuck prover process must cost this entry, not the line
    watchdog.daemon = True
    watchdog.start()
    try:
        for p in procs:
            line = p.stdout.readline()
            if line.strip() != "ready":
                return {"error": "a prover process did not come up"}
        for p in procs:
            p.stdin.write("go\n"); p.stdin.flush()
        outs = [json.loads(p.stdout.readline()) for p in procs]
    except Exception as e:                # noqa: BLE001
        return {"error": repr(e)[:200]}
    finally:
        watchdog.cancel()
        for p in procs:
            try:
                p.wait(timeout=30)
            except Exception:             # noqa: BLE001
                p.kill()
    return {"proofs_per_s": sum(o["proofs"] for o in outs) / max(o["seconds"] for o in outs), "sha256": sorted({o["sha256"] for o in outs})}


HELLO_WORLD = "++++++++[>++++[>++>+++>+++>+<<<<-]>+>+>->>+[<]<-]>>.>---.+++++++..+++.>>.<-.<.+++.------.--------.>>+.>++."


class FixedRandomness:
    """count -> bytes, the same sequence for every prover and every proof (a stand-in for os.urandom that makes proofs comparable)"""
    expand_on_device = True

    def __init__(self):
        import hashlib
        self.pos, self.buf = 0, hashlib.shake_256(b"bench-concurrent").digest(1 << 16)

    def __call__(self, count):
        out = self.buf[self.pos:self.pos + count]
        self.pos += count
        return out


def bench_stark_concurrent(ks=(1, 2, 4), seconds=0.8):
    """the same for the Hello-World proof: K threads, each with its own torch stream (the package enqueues on the thread's current stream)
    and its own BrainfuckStark instance, prove() back to back.  The prover's host side is Python, so the threads share the interpreter
    lock: what does not scale here is host time, not the GPU.  Randomness comes from one fixed byte stream per prover
    (randomness.override), so every prover's proof must be the same bytes."""
    import hashlib
    import torch
    from stark_brainfuck_amd import randomness
    from stark_brainfuck_amd.brainfuck_stark import BrainfuckStark
    from stark_brainfuck_amd.vm import VirtualMachine
    program = VirtualMachine.compile(HELLO_WORLD)
    running_time, inputs, outputs = VirtualMachine.run(program)
    Fixed = FixedRandomness
    out, digests = {}, set()
    for K in ks:
        streams = [torch.cuda.Stream() for _ in range(K)]
        mats = [VirtualMachine.simulate(program, input_data=inputs) for _ in range(K)]
        first = [None] * K

        def make(i):
            def one():
                with torch.cuda.stream(streams[i]), randomness.override(Fixed()):
                    stark = BrainfuckStark(running_time, len(mats[i][1]), program, inputs, outputs)
                    proof = stark.prove(program, *mats[i])
                if first[i] is None:
                    first[i] = hashlib.sha256(proof).hexdigest()
            return one
        r = _throughput([make(i) for i in range(K)], seconds)
        digests.update(first)
        out[str(K)] = round(r["proofs_per_s"], 1) if "proofs_per_s" in r else r
    base = out.get("1")
    # the same with K PROCESSES (one interpreter each): what the GPU can take when the host side is not serialised
    by_process = {}
    for K in (2, 4, 8):
        r = stark_process_throughput(K, seconds)
        by_process[str(K)] = round(r["proofs_per_s"], 1) if "proofs_per_s" in r else r
        if "sha256" in r:
            digests.update(r["sha256"])
    best = max([v for v in list(out.values()) + list(by_process.values()) if isinstance(v, float)] or [0.0])
    return {"proofs_per_s_by_provers": out, "proofs_per_s_by_prover_processes": by_process,
            "best_over_single": (best / base) if isinstance(base, float) and base else None,
            "proofs_identical_across_provers": len(digests) == 1,
            "note": "K threads x (own torch stream, own BrainfuckStark) share one interpreter lock; K processes do not; Hello World, prove() back to back for %.1f s per K" % seconds}


VALU_PER_COMPRESSION = 1983     # BLAKE2b-512 compression in VGPRs on gfx950: 801 xor + 576 funnel shifts + 575 64-bit adds + 31
SIMDS = 1024                    # 256 CUs x 4
NOMINAL_SCLK_MHZ = 2400


def valu_roofline(wave_instructions, seconds, what=None):
    """integer-VALU roofline of a piece of work: a wave64 VALU instruction occupies its SIMD for 4 cycles, so the chip retires at most
    1024 SIMDs x sclk / 4 of them per second.  `achieved` = the work's ALGORITHMIC wave instructions / its time; peak at the nominal
    2.4 GHz (a clock sampled during ANOTHER leg says nothing about this one: round-3 verdict, weak #5)."""
    achieved = wave_instructions / seconds / 1e9
    peak = SIMDS * NOMINAL_SCLK_MHZ * 1e6 / 4 / 1e9
    r = {"bound": "valu_int", "achieved": achieved, "peak": peak, "unit": "G wave-instructions/s", "frac": achieved / peak}
    if what:
        r["model"] = what
    return r


def fri_roofline(fri):
    """Fri.prove is BLAKE2b: round r commits to N_r = N / 2^r extension elements -- 3 compressions per leaf behind the tabulated
    first-block state (a 385..409-byte pickle is 4 blocks) and one per tree node -- so ~4 N_r compressions per round, 8 N in all;
    the fold is 21 multiplications per element next to ~6000 instructions of hashing.  Timed over `breakdown_ms.rounds` (the commit
    phase incl. its host round trips)."""
    N, rounds = fri["N"], fri["rounds"]
    compressions = sum(4 * (N >> r) - 1 for r in range(rounds))
    r = valu_roofline(compressions * VALU_PER_COMPRESSION / 64.0, fri["breakdown_ms"]["rounds"] * 1e-3,
                      "%d BLAKE2b compressions x %d VALU / 64 lanes, over breakdown_ms.rounds" % (compressions, VALU_PER_COMPRESSION))
    r["compressions"] = compressions
    r["hbm_frac"] = 376.0 * N / (fri["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
    return r


def bench_merkle_tree(lib, _lib, stream, log_n, steps=10):
    """Merkle(codeword) over 2^log_n extension elements (merkle.py:8-41, the commitment of an FRI round) on its own, HIP events on the
    kernels' stream: merkle_leaves_xfe_kernel + merkle_parents_kernel, the two kernels that are 90 % of Fri.prove at this size."""
    from stark_brainfuck_amd.device import DeviceBuffer
    n = 1 << log_n
    limbs = DeviceBuffer.from_numpy(felt_array(SEED + 77, 0, 3 * n))
    nodes = DeviceBuffer(2 * n * 8)

    def one():
        _lib.check(lib.bfs_merkle_build_xfe(limbs.ptr, n, n, nodes.ptr, stream))
    for _ in range(3):
        one()
    e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
    lib.bfs_event_create(ctypes.byref(e0)); lib.bfs_event_create(ctypes.byref(e1))
    lib.bfs_event_record(e0, stream)
    for _ in range(steps):
        one()
    lib.bfs_event_record(e1, stream)
    _lib.check(lib.bfs_stream_synchronize(stream))
    ms = ctypes.c_float()
    _lib.check(lib.bfs_event_elapsed_ms(e0, e1, ctypes.byref(ms)))
    per = ms.value / steps
    compressions = 4 * n - 1
    r = valu_roofline(compressions * VALU_PER_COMPRESSION / 64.0, per * 1e-3,
                      "(3 n leaf + n - 1 node) BLAKE2b compressions x %d VALU / 64 lanes" % VALU_PER_COMPRESSION)
    r["hbm_frac"] = (24.0 + 128.0) * n / (per * 1e-3) / 1e9 / HBM_PEAK_GBS
    limbs.free(); nodes.free()
    return {"ms": per, "leaves": n, "leaves_per_s": n / per * 1e3, "algorithmic_GBps": 152.0 * n / per / 1e6, "steps": steps, "roofline": r,
            "note": "bfs_merkle_build_xfe, HIP events over %d back-to-back trees" % steps}


def prover_kernel_table(workload):
    """the tracked per-kernel evidence (profiles/prover_valu.json, made by tools/prof_prover.sh + tools/make_prover_valu.py from rocprofv3
    --stats and --pmc runs): static, like roofline.traffic -- bench.py does not run a profiler"""
    path = os.path.join(ROOT, "profiles", "prover_valu.json")
    if not os.path.exists(path):
        return None
    w = json.load(open(path))["workloads"].get(workload)
    if not w:
        return None
    rows = {}
    for k, e in w["kernels"].items():
        if e["total_ms"] / max(e["calls"], 1) < 0.02 and e["total_ms"] < 0.5:
            continue
        rows[k] = {f: (round(v, 4) if isinstance(v, float) else v) for f, v in e.items()
                   if f in ("calls", "avg_us", "valu_wave_instructions_per_launch", "valu_issue_frac", "hbm_bytes_per_launch", "hbm_frac", "clock_hz_used")}
    return {"source": "static: profiles/prover_valu.json (rocprofv3 --kernel-trace --stats; --pmc SQ_INSTS_VALU / GRBM_GUI_ACTIVE / FETCH_SIZE / "
                      "WRITE_SIZE in separate passes; calls are over the 3 proofs (6 FRI runs) of the profiled command)", "kernels": rows}


def bench_fri(lib, _lib, stream, log_d):
    """config 3 (log_d = 18) and its 2^24 sibling: Fri.prove on a degree-2^log_d extension codeword, expansion 4,
    4 colinearity tests, fresh proof stream."""
    from stark_brainfuck_amd.device import DeviceBuffer
    expansion, t = 4, 4
    d, N = 1 << log_d, (1 << log_d) * expansion
    log_N = N.bit_length() - 1
    omega = lib.bfs_gl_primitive_root(log_N)
    coeffs = felt_array(SEED, 0, 3 * d).reshape(d, 3).T.copy()
    d_coef = DeviceBuffer.from_numpy(coeffs.reshape(-1))
    d_cw = DeviceBuffer(3 * N)
    _lib.check(lib.bfs_gl_ntt(d_coef.ptr, d, d, d_cw.ptr, N, log_N, 3, omega, 7, 1, stream))     # Domain.xevaluate
    _lib.check(lib.bfs_stream_synchronize(stream))
    times, idx = [], None
    for rep in range(6):
        ps = lib.bfs_ps_new()
        out = (ctypes.c_uint64 * t)()
        t0 = time.perf_counter()
        _lib.check(lib.bfs_fri_prove(ps, d_cw.ptr, N, log_N, 7, omega, expansion, t, out, stream))
        _lib.check(lib.bfs_stream_synchronize(stream))
        times.append(time.perf_counter() - t0)
        idx = [int(x) for x in out]
        tm = (ctypes.c_double * 6)()
        lib.bfs_fri_last_timing(tm)
        nobj = lib.bfs_ps_num_objects(ps)
        lib.bfs_ps_free(ps)
    ms = statistics.median(times[1:]) * 1e3
    return {"ms": ms, "N": N, "expansion": expansion, "colinearity_tests": t, "rounds": log_N - 2, "objects": nobj,
            "breakdown_ms": {k: round(v, 4) for k, v in zip(("rounds", "last_codeword", "fiat_shamir_sampling", "plan_openings", "gather", "build_objects"), tm)},
            "algorithmic_GBps": 376.0 * N / (ms * 1e-3) / 1e9,
            "note": "bfs_fri_prove through the C ABI, codeword resident in HBM, incl. host Fiat-Shamir round trips and D2H of openings"}


def bench_stark(code=None, label="Hello World!"):
    """config 4: BrainfuckStark.prove on the "Hello World!" program (FRI domain 2^17, 16 base + 10 extension columns, 52
    quotients), through the Python mirror of the reference's call surface; the proof is checked with verify().
    With another `code`: the same on that program (the 2^22-domain leg uses nested loops, 37 254 cycles)."""
    from stark_brainfuck_amd.brainfuck_stark import BrainfuckStark
    from stark_brainfuck_amd.vm import VirtualMachine
    if code is None:
        code = "++++++++[>++++[>++>+++>+++>+<<<<-]>+>+>->>+[<]<-]>>.>---.+++++++..+++.>>.<-.<.+++.------.--------.>>+.>++."
    program = VirtualMachine.compile(code)
    running_time, inputs, outputs = VirtualMachine.run(program)
    VirtualMachine.simulate(program, input_data=inputs)            # library load etc. outside the trace timing
    t0 = time.perf_counter()
    matrices = VirtualMachine.simulate(program, input_data=inputs)
    trace_ms = (time.perf_counter() - t0) * 1e3
    times, timing, proof = [], None, None
    rep = 0
    while rep < 4 or (rep < 24 and times[-1] < 0.008):       # a short proof is repeated more often: the median of 3 wobbled by 10 %
        stark = BrainfuckStark(running_time, len(matrices[1]), program, inputs, outputs)
        t0 = time.perf_counter()
        proof = stark.prove(program, *matrices)
        times.append(time.perf_counter() - t0)
        rep += 1
    # the stage breakdown comes from one more proof that synchronises its stream after every stage (the timed ones do not: their
    # stages overlap host and GPU work freely, so the breakdown adds up to slightly more than `ms`)
    stark = BrainfuckStark(running_time, len(matrices[1]), program, inputs, outputs)
    stark.stage_timing = True
    t0 = time.perf_counter()
    staged_proof = stark.prove(program, *matrices)
    staged_ms = (time.perf_counter() - t0) * 1e3
    timing = stark.timing
    assert len(staged_proof) > 0
    t0 = time.perf_counter()
    ok = BrainfuckStark(running_time, len(matrices[1]), program, inputs, outputs).verify(proof)
    verify_ms = (time.perf_counter() - t0) * 1e3
    busy = None
    bpath = os.path.join(ROOT, "profiles", "prove_busy.json")
    if code == HELLO_WORLD and os.path.exists(bpath):
        # GPU-busy share of the proof: the kernels + copies of one proof from the tracked rocprofv3 timeline (static, profiles/prove_busy.json)
        # over THIS run's wall clock -- what of the proof's latency is the device, the rest being host work and round trips in front of it
        rec = json.load(open(bpath))
        busy = {"gpu_busy_frac": min(1.0, rec["kernel_and_copy_busy_us"] * 1e-3 / (statistics.median(times[1:]) * 1e3)),
                "kernel_and_copy_busy_us_static": rec["kernel_and_copy_busy_us"], "static": "profiles/prove_busy.json"}
    return {"ms": statistics.median(times[1:]) * 1e3, "gpu_busy": busy, "program": label, "running_time": running_time,
            "fri_domain_length": stark.fri.domain.length, "proof_bytes": len(proof), "verified": bool(ok),
            "trace_ms": trace_ms, "verify_ms": verify_ms,
            "breakdown_ms": {k: round(v * 1e3, 2) for k, v in timing.items()}, "breakdown_run_ms": staged_ms,
            "note": "wall clock of prove() incl. host steps; reference: not runnable at this size (> 12 h extrapolated, BASELINE.md)"}


def stark_roofline(stark):
    """the two stages that dominate a large proof against the integer-VALU roofline: the zipped-row commitments (row_leaves_kernel:
    BLAKE2b over a pickled ROW per leaf) and the combination (air_combine_kernel<TABLE>: constraints + weighted sums at every point).
    Their VALU instruction counts are the tracked PMC figures (static); the time is this run's stage time."""
    tab = prover_kernel_table("stark22")
    if tab is None:
        return None
    k = tab["kernels"]
    b = stark["breakdown_ms"]

    def stage(names, ms, what):
        instr = 0.0
        for kk, e in k.items():
            if any(kk.startswith(name) for name in names) and "valu_wave_instructions_per_launch" in e:
                instr += e["valu_wave_instructions_per_launch"] * e["calls"] / 3.0      # the profiled command runs three proofs
        r = valu_roofline(instr, ms * 1e-3, what)
        r["stage_ms"] = ms
        return r
    return {"zipped_row_commitments": stage(["row_leaves_kernel", "row_leaves_generated_kernel", "row_pattern_kernel"], b["base_tree"] + b["ext_tree"],
                                            "static SQ_INSTS_VALU of the row-leaf kernels (profiles/prover_valu.json) over base_tree + ext_tree of this run"),
            "combination": stage(["air_combine_kernel", "zerofier_inverses_kernel", "difference_combine_kernel"], b["combination"],
                                 "static SQ_INSTS_VALU of air_combine<TABLE> + zerofier_inverses + difference_combine over this run's combination stage"),
            "static": "profiles/prover_valu.json"}


def bench_stark_cooperative(world, rank, device, dist, torch):
    """one proof of the 37 254-cycle program (FRI domain 2^22) by all ranks together: shard.shared_randomness + BrainfuckStark.cooperate"""
    from stark_brainfuck_amd import shard
    from stark_brainfuck_amd.brainfuck_stark import BrainfuckStark
    from stark_brainfuck_amd.vm import VirtualMachine
    program = VirtualMachine.compile("+" * 64 + "[>" + "+" * 64 + "[>++++<-]<-]+++.")
    running_time, inputs, outputs = VirtualMachine.run(program)
    matrices = VirtualMachine.simulate(program, input_data=inputs)
    times, proof = [], None
    for rep in range(3):
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with shard.shared_randomness(world, rank):
            stark = BrainfuckStark(running_time, len(matrices[1]), program, inputs, outputs).cooperate(world, rank, device=device)
            proof = stark.prove(program, *matrices)
        times.append(time.perf_counter() - t0)
    t = torch.tensor([min(times[1:])], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok = BrainfuckStark(running_time, len(matrices[1]), program, inputs, outputs).verify(proof) if rank == 0 else None
    return {"ms": float(t[0]) * 1e3, "ranks": world, "fri_domain_length": stark.fri.domain.length, "proof_bytes": len(proof), "verified": ok,
            "breakdown_ms": {k: round(v * 1e3, 2) for k, v in stark.timing.items()},
            "note": "every rank runs the transforms on all columns, hashes 1/N of the rows of the two zipped commitments and computes "
                    "quotients + combination for 1/N of the points; exchanged: 64-byte subtree roots, the opened paths, and one "
                    "all-gather of the combination codeword (24 bytes per point) before FRI"}


def cpu_baseline(log_n):
    """the oracle's C restatement of ntt.py (recursive radix-2, one core) on a bounded sample of the same workload."""
    from oracle import ref_oracle as o
    n = 1 << log_n
    sample_cols = 3 if log_n >= 24 else 4          # ~12 s on one host core at 2^24
    w = o.primitive_nth_root(n)
    t = 0.0
    for c in range(sample_cols):
        v = felt_array(SEED + (c << 32), 0, n)
        t0 = time.perf_counter()
        o.ntt(w, v)
        t += time.perf_counter() - t0
    return {"value": n * sample_cols / t, "unit": "elements/s", "cores": 1, "kind": "port",
            "sample": "%d column(s) of 2^%d elements, forward NTT, oracle/gl_oracle.c (gcc -O2), %.1f s" % (sample_cols, log_n, t),
            "host_cpus": os.cpu_count(),
            # the reference itself cannot travel to the GPU box; its own figure, measured in the build container (BASELINE.md section 2,
            # tests/golden/ntt20.json ref_seconds): CPython 3.10, 1 core, ntt.py on 2^20 elements in 242.7 s
            "reference_python": {"value": 4320.0, "unit": "elements/s", "cores": 1,
                                 "provenance": "BASELINE.md: ntt.py, n = 2^20, 242.7 s, build container"}}


def sample_clock_under_load(gpu_index, delay_s=3.0):
    """shader clock and package power as rocm-smi reports them `delay_s` into the sustained leg (best effort: None when rocm-smi is
    missing or prints something else).  The 8 x 2^24 step runs the package into its power limit, and the clock that results --
    not the nominal 2.4 GHz -- is what the VALU-issue bound of DESIGN.md 4.1 has to be priced at."""
    import re
    import subprocess
    time.sleep(delay_s)
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20).stdout
    except Exception:
        return None
    res = {}
    m = re.search(r"GPU\[%d\]\s*:\s*sclk clock level: \d+: \((\d+)Mhz\)" % gpu_index, out)
    if m:
        res["sclk_mhz_under_load"] = int(m.group(1))
    m = re.search(r"GPU\[%d\]\s*:\s*mclk clock level: \d+: \((\d+)Mhz\)" % gpu_index, out)
    if m:
        res["mclk_mhz_under_load"] = int(m.group(1))
    m = re.search(r"GPU\[%d\]\s*:.*Power \(W\): ([0-9.]+)" % gpu_index, out)
    if m:
        res["package_power_w_under_load"] = float(m.group(1))
    return res or None


def cpu_baseline_python():
    """the pure-Python CPU path, timed on this box (after the sustained GPU leg: a pure-Python loop holds the interpreter lock
    that the thread feeding the GPU needs between its calls)."""
    from oracle import ref_oracle as o
    # north_star asks for the pure-Python CPU path timed on this box: the reference cannot travel, so its cost model does --
    # oracle.ntt_python is ntt.py:4-23 on boxed elements (one object per value, arithmetic through the field object, per-index
    # square-and-multiply powers), pinned on the reference's goldens; here in the build container it runs 2^14 in 1.49 s against
    # 1.67 s for the reference itself (2^12: 0.30 s / 0.41 s).  Its rate falls with n (n log^2 n), so the size is part of the figure.
    py = []
    for lg in (14, 16):
        m = 1 << lg
        v = felt_array(SEED, 0, m).tolist()
        wm = o.primitive_nth_root(m)
        t0 = time.perf_counter()
        o.ntt_python(wm, v)
        dt = time.perf_counter() - t0
        py.append({"log_n": lg, "seconds": round(dt, 3), "elements_per_s": round(m / dt, 1)})
    return {"kind": "oracle.ntt_python (ntt.py:4-23 on boxed elements), 1 core, timed here", "runs": py}


if __name__ == "__main__":
    main()
