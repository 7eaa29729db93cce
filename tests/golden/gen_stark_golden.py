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

REF = os.environ.get("BFS_REFERENCE", "/root/reference/code")
sys.path.insert(0, REF)
sys.setrecursionlimit(100000)
HERE = os.path.dirname(os.path.abspath(__file__))


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
