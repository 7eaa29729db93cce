#!/usr/bin/env python3
"""Golden for the round-5 advisor's finding on terminals: runs the REFERENCE's BrainfuckStark.prove on the program '+' (no input: the
input evaluation terminal is zero) with `get_terminals` wrapped so that this terminal is the one-coefficient polynomial [p] -- the field's
zero stored as p -- and then the reference's own verify on the result.  Everything in the proof is consistent (the Fiat-Shamir hashes see
the same pickle on both sides, every arithmetic use of the terminal reduces); the reference nevertheless returns False, because its last
check compares the pulled terminal OBJECT with a computed element through Polynomial.__eq__ / BaseFieldElement.__eq__, i.e. the stored
coefficient values (brainfuck_stark.py:574-577, univariate.py:67-74, algebra.py:48-49).  A verifier that reduces the terminals it reads
accepts this proof; tests/test_stark_host.py pins both routes here to the reference's verdict.

Runs ONLY in the build container (imports /root/reference/code).  Writes tests/golden/noncanonical_terminal.json and _proof.bin.
"""
import sys
sys.dont_write_bytecode = True
import hashlib, json, os, time

REF = os.environ.get("BFS_REFERENCE", "/root/reference/code")
sys.path.insert(0, REF)
sys.setrecursionlimit(100000)
HERE = os.path.dirname(os.path.abspath(__file__))


class Stream:
    def __init__(self, tag):
        self.tag, self.pos, self.buf = tag, 0, b""

    def __call__(self, n):
        end = self.pos + n
        if end > len(self.buf):
            self.buf = hashlib.shake_256(b"bfs-golden-urandom" + self.tag).digest(max(2 * end, 1 << 16))
        out = self.buf[self.pos:end]
        self.pos = end
        return out


def main():
    stream = Stream(b"noncanonical-terminal")
    os.urandom = stream
    import salted_merkle
    salted_merkle.urandom = stream
    import brainfuck_stark as bs
    from algebra import BaseFieldElement
    from extension_field import ExtensionFieldElement
    from univariate import Polynomial
    from vm import VirtualMachine
    code = "+"
    program = VirtualMachine.compile(code)
    running_time, input_symbols, output_symbols = VirtualMachine.run(program, input_data=[])
    pm, mm, im, inm, om = VirtualMachine.simulate(program, input_data=[])
    rec = {"program": code, "input": ""}
    out = {}
    for tag in ("honest", "crafted"):
        stark = bs.BrainfuckStark(running_time, len(mm), program, input_symbols, output_symbols)
        if tag == "crafted":
            plain = stark.get_terminals
            p = stark.field.p

            def crafted_terminals():
                t = plain()
                assert t[2].is_zero(), "no input: the input evaluation terminal is zero"
                t[2] = ExtensionFieldElement(Polynomial([BaseFieldElement(p, stark.field)]), stark.xfield)
                return t
            stark.get_terminals = crafted_terminals
        t0 = time.time()
        pm, mm, im, inm, om = VirtualMachine.simulate(program, input_data=[])
        proof = stark.prove(program, pm, mm, im, inm, om)
        verdict = bool(bs.BrainfuckStark(running_time, len(mm), program, input_symbols, output_symbols).verify(proof))
        out[tag] = proof
        rec[tag] = {"proof_len": len(proof), "proof_sha256": hashlib.sha256(proof).hexdigest(), "reference_verify": verdict,
                    "prove_seconds": round(time.time() - t0, 1)}
        print(tag, rec[tag], flush=True)
    assert rec["honest"]["reference_verify"] is True and rec["crafted"]["reference_verify"] is False
    with open(os.path.join(HERE, "noncanonical_terminal.json"), "w") as f:
        json.dump(rec, f, indent=1)
    with open(os.path.join(HERE, "noncanonical_terminal_proof.bin"), "wb") as f:
        f.write(out["crafted"])
    with open(os.path.join(HERE, "noncanonical_terminal_honest_proof.bin"), "wb") as f:
        f.write(out["honest"])


if __name__ == "__main__":
    main()
