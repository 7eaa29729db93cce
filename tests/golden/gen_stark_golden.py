#!/usr/bin/env python3
"""Golden vectors for SURVEY.md 8f-1/8f-2: runs the *reference's* BrainfuckStark.prove (pure Python, imported from
/root/reference) on small Brainfuck programs with os.urandom replaced by a deterministic byte stream, and records what
a re-implementation has to reproduce stage by stage: table heights, FRI domain, SHA-256 of every base / extension /
quotient codeword, Merkle roots, challenges, terminals, degree bounds, opened indices, and the proof bytes themselves.

Runs ONLY in the build container (needs /root/reference).  Nothing of the reference's source is copied: the fixtures
are numbers, digests and the byte strings it produced.

    python tests/golden/gen_stark_golden.py <name> '<brainfuck program>' [input string]

Deterministic randomness: urandom(n) returns the next n bytes of SHAKE-256("bfs-golden-urandom" || name); the
re-implementation consumes the same stream through its injectable `urandom`.
"""
import sys
sys.dont_write_bytecode = True
import os, json, hashlib, struct, time
import multiprocessing          # (imported BEFORE os.urandom is replaced: the module draws its process authentication key from os.urandom on import)
_REAL_URANDOM = os.urandom

REF = os.environ.get("BFS_REFERENCE", "/root/reference/code")
sys.path.insert(0, REF)
sys.setrecursionlimit(100000)
HERE = os.environ.get("BFS_GOLDEN_OUT") or os.path.dirname(os.path.abspath(__file__))


class Stream:
    def __init__(self, tag):
        self.tag = tag
        self.pos = 0
        self.calls = []
        self.buf = b""

    def __call__(self, n):
        end = self.pos + n
        if end > len(self.buf):
            self.buf = hashlib.shake_256(b"bfs-golden-urandom" + self.tag).digest(max(2 * end, 1 << 16))
        out = self.buf[self.pos:end]
        self.pos = end
        self.calls.append(n)
        return out


def sha_elems(vals):
    """vals: list of base elements, extension elements or tuples thereof -> sha256 of little-endian u64 limbs (XFE as 3 limbs)."""
    h = hashlib.sha256()
    for v in vals:
        for e in (v if isinstance(v, tuple) else (v,)):
            if hasattr(e, "polynomial"):
                c = [x.value for x in e.polynomial.coefficients]
                c += [0] * (3 - len(c))
                h.update(struct.pack("<3Q", *c))
            else:
                h.update(struct.pack("<Q", e.value))
    return h.hexdigest()


def xl3(e):
    c = [x.value for x in e.polynomial.coefficients]
    return c + [0] * (3 - len(c))


def main():
    name, code = sys.argv[1], sys.argv[2]
    input_string = sys.argv[3] if len(sys.argv) > 3 else ""
    stream = Stream(name.encode())
    os.urandom = stream
    import salted_merkle
    salted_merkle.urandom = stream
    import table as table_mod
    import brainfuck_stark as bs
    import fri as fri_mod
    from vm import VirtualMachine
    from merkle import Merkle
    from ip import ProofStream

    rec = {"name": name, "program": code, "input": input_string, "python": sys.version.split()[0]}
    program = VirtualMachine.compile(code)
    running_time, input_symbols, output_symbols = VirtualMachine.run(program, input_data=list(input_string))
    pm, mm, im, inm, om = VirtualMachine.simulate(program, input_data=list(input_symbols))
    rec["compiled_program"] = [e.value for e in program]
    rec["running_time"] = running_time
    rec["output"] = "".join(output_symbols)
    rec["matrix_shapes"] = {"processor": [len(pm), 7], "memory": [len(mm), len(mm[0]) if mm else 0], "instruction": [len(im), 3],
                            "input": [len(inm), 1], "output": [len(om), 1]}
    rec["matrix_sha"] = {"processor": sha_elems([tuple(r) for r in pm]), "memory": sha_elems([tuple(r) for r in mm]),
                         "instruction": sha_elems([tuple(r) for r in im]), "input": sha_elems([tuple(r) for r in inm]),
                         "output": sha_elems([tuple(r) for r in om])}

    t0 = time.time()
    stark = bs.BrainfuckStark(running_time, len(mm), program, input_symbols, output_symbols)
    rec["setup_seconds"] = time.time() - t0
    rec["max_degree"] = stark.max_degree
    rec["fri_domain_length"] = stark.fri.domain.length
    rec["expansion_factor"] = stark.expansion_factor
    rec["num_colinearity_checks"] = stark.num_colinearity_checks
    rec["security_level"] = stark.security_level
    rec["table_heights"] = [t.height for t in stark.tables]
    rec["table_lengths"] = [t.length for t in stark.tables]

    # ---- instrumentation of prove(): wrap the collaborators, never the arithmetic
    cap = {"salted": [], "merkle": [], "quotients": [], "fs": []}
    orig_salted = bs.SaltedMerkle

    class CapSalted(orig_salted):
        def __init__(self, data_array):
            super().__init__(data_array)
            cap["salted"].append({"width": len(data_array[0]), "n": len(data_array), "root": self.root().hex(),
                                  "columns_sha": [sha_elems([row[j] for row in data_array]) for j in range(len(data_array[0]))],
                                  "salt0": self.leafs[0][1].hex(), "salt_last": self.leafs[-1][1].hex()})
    bs.SaltedMerkle = CapSalted
    orig_merkle = bs.Merkle

    class CapMerkle(orig_merkle):
        def __init__(self, data_array):
            super().__init__(data_array)
            cap["merkle"].append({"n": len(data_array), "root": self.root().hex(), "sha": sha_elems(data_array)})
    bs.Merkle = CapMerkle
    orig_allq = table_mod.Table.all_quotients

    def cap_allq(self, domain, codewords, challenges, terminals):
        q = orig_allq(self, domain, codewords, challenges, terminals)
        cap["quotients"].append({"table": type(self).__name__, "count": len(q), "sha": [sha_elems(c) for c in q],
                                 "degree_bounds": self.all_quotient_degree_bounds(challenges, terminals),
                                 "challenges": [xl3(c) for c in challenges], "terminals": [xl3(t) for t in terminals]})
        return q
    table_mod.Table.all_quotients = cap_allq
    orig_pq = bs.PermutationArgument.quotient

    def cap_pq(self, fri_domain):
        q = orig_pq(self, fri_domain)
        cap.setdefault("perm_quotients", []).append({"sha": sha_elems(q), "degree_bound": self.quotient_degree_bound()})
        return q
    bs.PermutationArgument.quotient = cap_pq
    orig_friprove = fri_mod.Fri.prove

    def cap_friprove(self, codeword, proof_stream):
        cap["objects_before_fri"] = len(proof_stream.objects)
        idx = orig_friprove(self, codeword, proof_stream)
        cap["fri_indices"] = idx
        return idx
    bs.Fri.prove = cap_friprove
    orig_sample_weights = bs.BrainfuckStark.sample_weights

    def cap_sw(self, number, randomness):
        w = orig_sample_weights(self, number, randomness)
        cap["fs"].append({"number": number, "seed": randomness.hex(), "first": xl3(w[0]), "last": xl3(w[-1])})
        return w
    bs.BrainfuckStark.sample_weights = cap_sw
    orig_si = bs.BrainfuckStark.sample_indices

    def cap_si(number, randomness, bound):
        r = orig_si(number, randomness, bound)
        cap["indices"] = r
        return r
    bs.BrainfuckStark.sample_indices = staticmethod(cap_si)

    workers = int(os.environ.get("BFS_GOLDEN_WORKERS", "1"))
    if workers > 1:
        install_parallel_quotients(table_mod, workers, rec)
    if os.environ.get("BFS_GOLDEN_CKPT"):
        install_early_cache(os.environ["BFS_GOLDEN_CKPT"], name, (stark, program, pm, mm, im, inm, om, VirtualMachine.field), table_mod, fri_mod, rec)
    ps = ProofStream()
    t0 = time.time()
    proof = stark.prove(program, pm, mm, im, inm, om, proof_stream=ps)
    rec["prove_seconds"] = time.time() - t0
    rec["urandom_calls"] = len(stream.calls)
    rec["urandom_bytes"] = stream.pos
    rec["urandom_call_sizes_rle"] = rle(stream.calls)
    rec["base_tree"], rec["extension_tree"] = cap["salted"][0], cap["salted"][1]
    rec["combination_tree"] = cap["merkle"][0]
    rec["quotients"] = cap["quotients"]
    rec["perm_quotients"] = cap.get("perm_quotients", [])
    rec["fiat_shamir"] = cap["fs"]
    rec["indices"] = cap["indices"]
    rec["fri_indices"] = cap["fri_indices"]
    rec["objects_before_fri"] = cap["objects_before_fri"]
    rec["num_objects"] = len(ps.objects)
    rec["terminals"] = [xl3(t) for t in stark.get_terminals()]
    rec["unit_distances"] = [t.unit_distance(stark.fri.domain.length) for t in stark.tables]
    rec["proof_len"] = len(proof)
    rec["proof_sha256"] = hashlib.sha256(proof).hexdigest()
    t0 = time.time()
    rec["verify"] = bool(stark.verify(proof))
    rec["verify_seconds"] = time.time() - t0
    with open(os.path.join(HERE, "stark_%s.json" % name), "w") as f:
        json.dump(rec, f, indent=1)
    if len(proof) <= (1 << 20):
        with open(os.path.join(HERE, "stark_%s_proof.bin" % name), "wb") as f:
            f.write(proof)
    print(json.dumps({k: rec[k] for k in ("name", "running_time", "max_degree", "fri_domain_length", "table_heights", "prove_seconds",
                                          "verify", "proof_len", "urandom_calls")}))


# ---- BFS_GOLDEN_WORKERS=N: the reference's quotient stage on N processes ------------------------------------------------------------
# Nine tenths of the reference's proving time is Table.all_quotients (table.py:155-301): for every constraint and every point of the FRI
# domain one MPolynomial.evaluate on boxed elements -- a pure function of the codewords, the challenges and the terminals (no randomness,
# nothing else reads its intermediate state).  A proof at FRI domain 2^16 takes a CPython process more than eight hours that way, one at
# 2^17 (the reference's own Hello-World test, test_brainfuck_stark.py:165-222) twice that.  With N > 1 this script forks N workers at each
# table's boundary / transition / terminal quotient call.  EVERY worker runs the reference's UNMODIFIED method over all constraints and
# all points; what differs is that the constraint objects it iterates over are gates: a gate forwards `evaluate(point)` to the reference's
# MPolynomial for the (constraint, point) pairs of the worker's share and answers the field's zero for the rest, so each worker pays
# for its share only.  The parent assembles every quotient codeword from the shares.  No arithmetic of the reference is restated or
# replaced, and the result is checked where it can be: `gen_stark_golden.py plus1 ...` with BFS_GOLDEN_WORKERS=4 writes the same
# stark_plus1_proof.bin, byte for byte, and the same digests of every quotient codeword, as the sequential run did (tests/golden/README).
_JOB = None


class _Gate:
    """stands where a constraint (MPolynomial) stands in the list a quotient method iterates over"""

    def __init__(self, mpo, ranges, calls_per_point, zero):
        self.mpo, self.ranges, self.calls_per_point, self.zero, self.calls = mpo, ranges, calls_per_point, zero, 0

    def evaluate(self, point):
        i = self.calls // self.calls_per_point
        self.calls += 1
        for a, b in self.ranges:
            if a <= i < b:
                return self.mpo.evaluate(point)
        return self.zero

    def __getattr__(self, name):          # anything else a method asks of a constraint goes to the constraint
        return getattr(self.mpo, name)


def _quotient_worker(share):
    """in a forked child: the reference's method on gated constraints; returns {(l, a, b): values}"""
    self, method_name, original, args, constraints_attr, constraints, calls_per_point, zero = _JOB
    mine = {}
    for l, a, b in share:
        mine.setdefault(l, []).append((a, b))
    gates = [_Gate(mpo, mine.get(l, []), calls_per_point, zero) for l, mpo in enumerate(constraints)]
    setattr(self, constraints_attr, lambda *unused: gates)          # the instance attribute shadows the class's method in this process only
    codewords = original(self, *args)
    return {(l, a, b): codewords[l][a:b] for l, a, b in share}


def install_parallel_quotients(table_mod, workers, rec):
    ctx = multiprocessing.get_context("fork")
    log = rec.setdefault("parallel_quotients", {"workers": workers, "calls": []})

    def wrap(method_name, constraints_attr, calls_per_point, constraint_args):
        original = getattr(table_mod.Table, method_name)

        def parallel(self, domain, codewords, challenges, *rest):
            global _JOB
            constraints = getattr(self, constraints_attr)(*constraint_args(challenges, rest))
            n = domain.length
            weight = [max(1, len(c.dictionary)) for c in constraints]
            if n < 64 or not constraints or self.height == 0:
                return original(self, domain, codewords, challenges, *rest)
            # pieces of roughly equal cost (terms x points).  The work goes out in ROUNDS of about ten minutes per worker; after every
            # round the partial result is pickled (BFS_GOLDEN_CKPT=<dir>), so that a run that was interrupted and is started again with
            # the same name recomputes only the cheap early stages (deterministic: same randomness stream) and the round it was in.  The
            # worker count is read again in front of every round (BFS_GOLDEN_WORKERS_FILE: a file holding the number), so that cores
            # freed by another run can be put to use without a restart.  The checkpoint's name carries a digest of the call's inputs.
            total = sum(weight) * n
            round_cost = int(os.environ.get("BFS_GOLDEN_ROUND_COST", "500000"))       # ~1.1 ms of CPython per term and point
            piece = max(64, min(total // (workers * 6), round_cost // 4))
            pieces = []
            for l, w in enumerate(weight):
                step = max(16, min(n, piece // w))
                for a in range(0, n, step):
                    pieces.append((w * (min(n, a + step) - a), l, a, min(n, a + step)))
            pieces.sort(reverse=True)
            ckpt = os.environ.get("BFS_GOLDEN_CKPT")
            ckpt_file = None
            out = [[None] * n for _ in constraints]
            done = set()
            import pickle
            if ckpt:
                key = hashlib.sha256(repr((rec["name"], type(self).__name__, method_name, n, sha_elems(codewords[0]), sha_elems(codewords[-1]),
                                           [xl3(c) for c in challenges], [xl3(t) for r in rest for t in r])).encode()).hexdigest()[:16]
                ckpt_file = os.path.join(ckpt, "%s_%s_%s_%s.pkl" % (rec["name"], type(self).__name__, method_name, key))
                if os.path.exists(ckpt_file):
                    with open(ckpt_file, "rb") as f:
                        done, out = pickle.load(f)
                    print("[parallel] %s.%s: %d of %d pieces from checkpoint %s" % (type(self).__name__, method_name, len(done), len(pieces), ckpt_file),
                          file=sys.stderr, flush=True)
            t0 = time.time()
            todo = [p for p in pieces if p[1:] not in done]
            rounds = 0
            while todo:
                wf = os.environ.get("BFS_GOLDEN_WORKERS_FILE")
                now = workers
                if wf and os.path.exists(wf):
                    try:
                        now = max(1, int(open(wf).read().split()[0]))
                    except (ValueError, IndexError):
                        now = workers
                shares, load = [[] for _ in range(now)], [0] * now
                rest_todo = []
                for cost, l, a, b in todo:
                    k = load.index(min(load))
                    if load[k] >= round_cost:
                        rest_todo.append((cost, l, a, b))
                        continue
                    shares[k].append((l, a, b))
                    load[k] += cost
                todo = rest_todo
                shares = [sh for sh in shares if sh]
                _JOB = (self, method_name, original, (domain, codewords, challenges) + tuple(rest), constraints_attr, constraints, calls_per_point, self.field.zero())
                stream_urandom, os.urandom = os.urandom, _REAL_URANDOM          # the pool's own needs must not draw from the proof's randomness
                try:
                    with ctx.Pool(len(shares)) as pool:
                        parts = pool.map(_quotient_worker, shares, chunksize=1)
                finally:
                    os.urandom = stream_urandom
                _JOB = None
                for part in parts:
                    for (l, a, b), values in part.items():
                        out[l][a:b] = values
                        done.add((l, a, b))
                rounds += 1
                if ckpt_file:
                    os.makedirs(ckpt, exist_ok=True)
                    with open(ckpt_file + ".tmp", "wb") as f:
                        pickle.dump((done, out), f, protocol=4)
                    os.replace(ckpt_file + ".tmp", ckpt_file)
                print("[parallel] %s.%s: round %d on %d workers done, %d of %d pieces, %.0f s so far" % (type(self).__name__, method_name, rounds, len(shares),
                                                                                                          len(done), len(pieces), time.time() - t0), file=sys.stderr, flush=True)
            assert all(v is not None for cw in out for v in cw)
            log["calls"].append({"table": type(self).__name__, "method": method_name, "constraints": len(constraints), "points": n,
                                 "seconds": round(time.time() - t0, 1), "pieces": len(pieces)})
            print("[parallel] %s.%s: %d constraints x %d points on %d workers, %.1f s" % (type(self).__name__, method_name, len(constraints), n, workers, time.time() - t0),
                  file=sys.stderr, flush=True)
            return out
        setattr(table_mod.Table, method_name, parallel)

    wrap("boundary_quotients", "boundary_constraints_ext", 1, lambda challenges, rest: (challenges,))
    wrap("transition_quotients", "transition_constraints_ext", 2, lambda challenges, rest: (challenges,))
    wrap("terminal_quotients", "terminal_constraints_ext", 1, lambda challenges, rest: (challenges,) + tuple(rest))


# ---- BFS_GOLDEN_CKPT=<dir>: the stages in front of the quotients survive an interruption too -------------------------------------------
# At FRI domain 2^16 the reference spends over two hours between the start of prove() and its first quotient: `fast_interpolate` of every
# column (table.py:112-136) and the coset evaluations `Domain.evaluate` / `xevaluate` (fri.py:26-37) -- pure functions of their arguments,
# called in a fixed order; the randomness is drawn OUTSIDE them.  With a checkpoint directory each call's result is pickled under its
# sequence number, and a run that is started again returns the stored result instead of computing it.  The proof is `pickle.dumps` of
# objects and pickle memoises by IDENTITY, so a stored result must come back referring to the SAME field objects the live computation would
# have used: the BaseField / ExtensionField instances reachable from the prover, the program and the trace are numbered by one
# deterministic traversal and written as persistent ids.  Checked like the parallel quotients: a run of `plus1` resumed from such
# checkpoints writes the committed stark_plus1_proof.bin byte for byte.
class _FieldRegistry:
    def __init__(self, roots):
        import io, pickle
        from algebra import BaseField
        from extension_field import ExtensionField
        self.pickle, self.io = pickle, io
        self.objects, seen = [], set()
        registry = self

        class Scan(pickle.Pickler):
            def persistent_id(self, obj):
                if isinstance(obj, (BaseField, ExtensionField)) and id(obj) not in seen:
                    seen.add(id(obj))
                    registry.objects.append(obj)
                return None
        Scan(io.BytesIO(), protocol=4).dump(roots)
        self.index = {id(o): i for i, o in enumerate(self.objects)}

    def dump(self, obj, path):
        registry = self

        class Writer(self.pickle.Pickler):
            def persistent_id(self, o):
                return registry.index.get(id(o))
        with open(path + ".tmp", "wb") as f:
            Writer(f, protocol=4).dump(obj)
        os.replace(path + ".tmp", path)

    def load(self, path):
        registry = self

        class Reader(self.pickle.Unpickler):
            def persistent_load(self, pid):
                return registry.objects[pid]
        with open(path, "rb") as f:
            return Reader(f).load()


def install_early_cache(directory, name, roots, table_mod, fri_mod, rec):
    os.makedirs(directory, exist_ok=True)
    registry = _FieldRegistry(roots)
    state = {"seq": 0}
    log = rec.setdefault("early_stage_checkpoints", {"fields_numbered": len(registry.objects), "computed": 0, "loaded": 0})

    # BFS_GOLDEN_EARLY_SHARE="h/H": H processes of this script run side by side on the same checkpoint directory; process h computes the
    # calls whose sequence number is h modulo H and WAITS for the files of the others (the columns of a table are interpolated and evaluated
    # independently of each other, so the H processes advance together through a table).  Process 0 carries on into the quotients; the
    # helpers (h > 0) stop at the first of them.
    share = os.environ.get("BFS_GOLDEN_EARLY_SHARE", "0/1").split("/")
    mine, parties = int(share[0]), int(share[1])
    if mine > 0:
        def helper_is_done(*unused_args, **unused_kwargs):
            print("[early] helper %d/%d: the early stages are on disk" % (mine, parties), file=sys.stderr, flush=True)
            os._exit(0)
        table_mod.Table.all_quotients = helper_is_done

    class _Pending:
        """the result of a call another process owns, not needed yet: inside Table.lde / ldex the interpolants and codewords of a table's
        columns are only collected in lists, so a process can run on to the calls it owns and pick the others' results up at the end"""
        def __init__(self, path):
            self.path, self.value, self.loaded = path, None, False

        def resolve(self):
            if not self.loaded:
                waited = time.time()
                while not os.path.exists(self.path):
                    time.sleep(1.0)
                    if time.time() - waited > 8 * 3600:
                        raise RuntimeError("no other process wrote %s" % self.path)
                self.value, self.loaded = registry.load(self.path), True
                log["loaded"] += 1
            return self.value

    def settle(seq):
        for k, item in enumerate(seq):
            if isinstance(item, _Pending):
                seq[k] = item.resolve()

    def cached(label, original):
        def call(*args, **kwargs):
            state["seq"] += 1
            path = os.path.join(directory, "%s_early_%04d_%s.pkl" % (name, state["seq"], label))
            if os.path.exists(path):
                log["loaded"] += 1
                return registry.load(path)
            if state["seq"] % parties != mine:
                pending = _Pending(path)
                return pending if state.get("collecting", 0) > 0 else pending.resolve()
            args = tuple(a.resolve() if isinstance(a, _Pending) else a for a in args)
            t0 = time.time()
            out = original(*args, **kwargs)
            registry.dump(out, path)
            log["computed"] += 1
            print("[early] %s #%d computed in %.0f s" % (label, state["seq"], time.time() - t0), file=sys.stderr, flush=True)
            return out
        return call
    if parties > 1:
        plain_lde, plain_ldex = table_mod.Table.lde, table_mod.Table.ldex

        def lde(self, domain):
            state["collecting"] = state.get("collecting", 0) + 1
            try:
                out = plain_lde(self, domain)
            finally:
                state["collecting"] -= 1
            settle(self.codewords)                    # (table.py:138-141: `out` IS self.codewords)
            if out is not self.codewords:
                settle(out)
            return out

        def ldex(self, domain, xfield):
            state["collecting"] = state.get("collecting", 0) + 1
            try:
                out = plain_ldex(self, domain, xfield)
            finally:
                state["collecting"] -= 1
            settle(out)                               # the same _Pending objects sit in both lists: each resolves once, to one object
            settle(self.codewords)
            return out
        table_mod.Table.lde, table_mod.Table.ldex = lde, ldex
    table_mod.fast_interpolate = cached("fast_interpolate", table_mod.fast_interpolate)
    fri_mod.Fri.Domain.evaluate = cached("evaluate", fri_mod.Fri.Domain.evaluate)
    fri_mod.Fri.Domain.xevaluate = cached("xevaluate", fri_mod.Fri.Domain.xevaluate)


def rle(xs):
    out = []
    for x in xs:
        if out and out[-1][0] == x:
            out[-1][1] += 1
        else:
            out.append([x, 1])
    return out


if __name__ == "__main__":
    main()
