#!/usr/bin/env python3
"""Golden for the DEBUG degree checks (round-5 verdict, next #6): runs the REFERENCE's BrainfuckStark.prove with DEBUG=1 on small traces
-- once as the VM wrote them, then with one corrupted cell each -- and records where its assertions stop it: the function of table.py /
brainfuck_stark.py, the table class, the index of the constraint (the loop variable of the frame that raised).  The re-implementation's
DEBUG mode (stark_brainfuck_amd/debug_checks.py) must pass / stop at the same places.

Runs ONLY in the build container (imports /root/reference/code); writes tests/golden/debug_checks.json.

    python tests/golden/gen_debug_golden.py
"""
import sys
sys.dont_write_bytecode = True
import contextlib, hashlib, io, json, os, time, traceback

REF = os.environ.get("BFS_REFERENCE", "/root/reference/code")
sys.path.insert(0, REF)
sys.setrecursionlimit(100000)
HERE = os.path.dirname(os.path.abspath(__file__))

PROGRAM = "++."          # 4 cycles, FRI domain 512: about three minutes of CPython per case
CASES = [
    {"tag": "clean", "matrix": None},
    {"tag": "processor_cell", "matrix": "processor", "row": 2, "column": 5, "add": 1},        # the memory value in the middle of the trace
    {"tag": "processor_first_row", "matrix": "processor", "row": 0, "column": 0, "add": 1},   # the cycle counter starts at 1
    {"tag": "instruction_cell", "matrix": "instruction", "row": 2, "column": 1, "add": 1},
    {"tag": "memory_cell", "matrix": "memory", "row": 1, "column": 2, "add": 5},
]


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
    program_text = sys.argv[1] if len(sys.argv) > 1 else PROGRAM
    os.environ["DEBUG"] = "1"
    import salted_merkle
    import brainfuck_stark as bs
    from vm import VirtualMachine
    rec = {"program": program_text, "cases": []}
    for case in CASES:
        stream = Stream(("debug-" + case["tag"]).encode())
        os.urandom = stream
        salted_merkle.urandom = stream
        program = VirtualMachine.compile(program_text)
        running_time, input_symbols, output_symbols = VirtualMachine.run(program, input_data=[])
        matrices = dict(zip(("processor", "memory", "instruction", "input", "output"), VirtualMachine.simulate(program, input_data=list(input_symbols))))
        rec.setdefault("shapes", {k: [len(m), len(m[0]) if m else 0] for k, m in matrices.items()})
        if case["matrix"]:
            m = matrices[case["matrix"]]
            cell = m[case["row"]][case["column"]]
            m[case["row"]][case["column"]] = cell + type(cell)(case["add"], cell.field)
        stark = bs.BrainfuckStark(running_time, len(matrices["memory"]), program, input_symbols, output_symbols)
        rec["fri_domain_length"] = stark.fri.domain.length
        t0 = time.time()
        out = {"tag": case["tag"], **{k: case.get(k) for k in ("matrix", "row", "column", "add")}}
        sink = io.StringIO()
        try:
            with contextlib.redirect_stdout(sink):
                proof = stark.prove(program, matrices["processor"], matrices["memory"], matrices["instruction"], matrices["input"], matrices["output"])
            out["outcome"] = "passed"
            out["proof_sha256"], out["proof_len"], out["urandom_bytes"] = hashlib.sha256(proof).hexdigest(), len(proof), stream.pos
        except (AssertionError, AttributeError) as e:
            # (AttributeError: the dump that table.py:219-234 prints in front of its assert(False) reads `self.terminal_index`, which the
            #  processor and instruction tables do not have -- the reference stops there, one line before the assertion it was heading for)
            frames = traceback.extract_tb(e.__traceback__)
            tb = e.__traceback__
            while tb.tb_next is not None:
                tb = tb.tb_next
            frame = tb.tb_frame
            loc = frame.f_locals
            out["outcome"] = "assertion" if isinstance(e, AssertionError) else "stopped in the failure dump in front of assert(False)"
            out["exception"] = type(e).__name__
            out["message"] = str(e)
            out["function"] = frame.f_code.co_name
            out["file"] = os.path.basename(frame.f_code.co_filename)
            out["line"] = tb.tb_lineno
            if "self" in loc:
                out["table"] = type(loc["self"]).__name__
            for name in ("l", "i"):
                if name in loc and isinstance(loc[name], int):
                    out["index_" + name] = loc[name]
            if "qc" in loc and "quotient_codewords" in loc:          # table.py:170-176 walks the codewords themselves: which one it was at
                out["index_qc"] = next(k for k, cw in enumerate(loc["quotient_codewords"]) if cw is loc["qc"])
            out["stack"] = [f.name for f in frames]
        out["seconds"] = round(time.time() - t0, 1)
        out["debug_lines_printed"] = sink.getvalue().count("\n")
        rec["cases"].append(out)
        print(json.dumps(out), flush=True)
    with open(os.path.join(HERE, "debug_checks.json"), "w") as f:
        json.dump(rec, f, indent=1)


if __name__ == "__main__":
    main()
