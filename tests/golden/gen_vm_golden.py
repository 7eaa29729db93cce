#!/usr/bin/env python3
"""Golden vectors for the caller side of the hot path: the *reference's* VirtualMachine (vm.py:172-306), Table.pad
and Table.extend (processor_table.py:24-31 / 359-427, instruction_table.py, memory_table.py, io_table.py) run on many small programs --
hand-picked edge cases (loops skipped on a zero cell, wraps below zero of a cell and of the memory pointer, nested loops, input and
output inside loops, empty input / output tables) and seeded random programs.  Recorded per program: compiled words, running time,
output, and SHA-256 digests of the five trace matrices, of the five padded tables and of the five extended tables under fixed
challenges / initials, plus the terminals.  Whole proofs of the reference take minutes to hours (gen_stark_golden.py: eight programs);
these stages take milliseconds, so they can be pinned on a much wider set of programs.

Runs ONLY in the build container (needs /root/reference).  Nothing of the reference's source is copied: the fixture is programs, numbers
and digests.

    python tests/golden/gen_vm_golden.py            ->  tests/golden/vm.json
"""
import sys
sys.dont_write_bytecode = True
import os, json, hashlib, struct, random

REF = os.environ.get("BFS_REFERENCE", "/root/reference/code")
sys.path.insert(0, REF)
sys.setrecursionlimit(100000)
HERE = os.path.dirname(os.path.abspath(__file__))

P = (1 << 64) - (1 << 32) + 1


def limbs(e):
    if hasattr(e, "polynomial"):
        c = [x.value for x in e.polynomial.coefficients]
        return c + [0] * (3 - len(c))
    return [e.value]


def sha_matrix(matrix, lift=False):
    """rows of base or extension elements -> sha256 of little-endian u64 limbs, row-major; lift: base elements as (v, 0, 0)"""
    h = hashlib.sha256()
    for row in matrix:
        for e in row:
            l = limbs(e)
            if lift and len(l) == 1:
                l = l + [0, 0]
            h.update(struct.pack("<%dQ" % len(l), *l))
    return h.hexdigest()


def random_program(rng, length):
    """a random well-bracketed program that terminates quickly: the generator tracks the machine while it writes, so `-` is only
    emitted on a cell that stays >= 0 and `<` only right of cell 0; loops are [-], [->+<] (entered) or start on a zero cell (skipped)"""
    out, inputs = [], []
    cells, mp = {}, 0
    while len(out) < length:
        k = rng.randrange(11)
        if k < 4:
            count = rng.randrange(1, 4)
            if rng.randrange(2) and cells.get(mp, 0) >= count:
                out.append("-" * count); cells[mp] = cells.get(mp, 0) - count
            else:
                out.append("+" * count); cells[mp] = cells.get(mp, 0) + count
        elif k < 6:
            if rng.randrange(2) and mp > 0:
                out.append("<"); mp -= 1
            else:
                out.append(">"); mp += 1
        elif k == 6:
            ch = rng.randrange(1, 12)
            out.append(","); inputs.append(chr(ch)); cells[mp] = ch
        elif k == 7:
            out.append(".")
        elif k == 8:
            if cells.get(mp, 0) > 0:
                out.append("[-]"); cells[mp] = 0
            else:
                out.append("[+>+<]")              # skipped: the cell is zero
        elif k == 9:
            v = cells.get(mp, 0)
            if 0 < v <= 12:
                out.append("[->+<]"); cells[mp + 1] = cells.get(mp + 1, 0) + v; cells[mp] = 0
            else:
                out.append("+")
                cells[mp] = v + 1
        else:
            v = cells.get(mp, 0)
            if 0 < v <= 4 and cells.get(mp + 1, 0) == 0:
                out.append("[>++[>+<-]<-]")       # nested
                cells[mp + 2] = cells.get(mp + 2, 0) + 2 * v; cells[mp] = 0
            else:
                out.append(">")
                mp += 1
    return "".join(out), "".join(inputs)


HAND_PICKED = [
    ("+", ""), ("-", ""), (">", ""), ("<", ""), (".", ""), (",", "a"), (",.", "\x00"), (",+.", "\xff"),
    ("[+].", ""), ("[[+]+]+.", ""), ("+[-]", ""), ("+>[-]<.", ""), ("+[>[+]<-]>.", ""),
    ("-<-.", ""), ("<+.>-.", ""), ("->-<[+].", ""),
    ("++[>++[>+<-]<-]>>.", ""), ("+++[>+++[>+++[>+<-]<-]<-]>>>.", ""),
    (",[.,]", "ab\x00"), (",[.-]", "\x05"), ("+[>,.<-]", "z"), ("++[>,.<-]", "xy"), (",>,<[->+<]>.", "!#"),
    ("+.+.+.+.+.", ""), (",,,,", "abcd"), (",.,.,.", "abc"), ("....", ""),
    ("++++++++[>++++++++<-]>+.", ""), (">>>+<<<+[>]<.", ""), ("+>+>+<<[>]<.", ""),
    ("++>+++<[->[->+>+<<]>>[-<<+>>]<<<]>>.", ""),
]


def main():
    import brainfuck_stark as bs
    from vm import VirtualMachine
    from extension_field import ExtensionField, ExtensionFieldElement
    from univariate import Polynomial
    from algebra import BaseFieldElement

    rng = random.Random(0xB7A1F)
    programs = list(HAND_PICKED) + [random_program(rng, rng.randrange(1, 28)) for _ in range(60)]
    records = []
    for code, inp in programs:
        program = VirtualMachine.compile(code)
        try:
            running_time, input_symbols, output_symbols = VirtualMachine.run(program, input_data=list(inp))
        except Exception as e:                        # e.g. `.` or `[` on a cell nothing was written to: KeyError in vm.py's run()
            records.append({"program": code, "input": inp, "compiled_program": [w.value for w in program], "run_error": type(e).__name__})
            print("%-50r run() raises %s" % (code[:48], type(e).__name__), flush=True)
            continue
        assert running_time < 3000, (code, running_time)
        pm, mm, im, inm, om = VirtualMachine.simulate(program, input_data=list(input_symbols))
        rec = {"program": code, "input": inp, "compiled_program": [e.value for e in program], "running_time": running_time,
               "input_symbols": "".join(input_symbols), "output": "".join(output_symbols),
               "matrix_lengths": [len(pm), len(mm), len(im), len(inm), len(om)],
               "matrix_sha": [sha_matrix(m) for m in (pm, mm, im, inm, om)]}
        stark = bs.BrainfuckStark(running_time, len(mm), program, input_symbols, output_symbols)
        rec["max_degree"] = stark.max_degree
        rec["fri_domain_length"] = stark.fri.domain.length
        rec["table_heights"] = [t.height for t in stark.tables]
        # tables in the prover's order: processor, instruction, memory, input, output (brainfuck_stark.py:63-64)
        for table, matrix in zip(stark.tables, (pm, im, mm, inm, om)):
            table.matrix = matrix
            table.pad()
        rec["padded_sha"] = [sha_matrix(t.matrix) for t in stark.tables]
        rec["padded_lengths"] = [len(t.matrix) for t in stark.tables]
        field = stark.field
        xfield = stark.xfield
        stream = hashlib.shake_256(b"bfs-golden-vm" + code.encode() + b"|" + inp.encode("latin-1", "replace")).digest(13 * 24)

        def xfe(k):
            ws = struct.unpack("<3Q", stream[24 * k:24 * k + 24])
            return ExtensionFieldElement(Polynomial([BaseFieldElement(w % P, field) for w in ws]), xfield)
        challenges = [xfe(k) for k in range(11)]
        initials = [xfe(11), xfe(12)]
        rec["challenges"] = [limbs(c) for c in challenges]
        rec["initials"] = [limbs(c) for c in initials]
        for table in stark.tables:
            table.codewords = []                      # extend() lifts the codewords lde() left behind; there are none here
            table.extend(challenges, initials)
        rec["extended_sha"] = [sha_matrix(t.matrix, lift=True) for t in stark.tables]
        rec["extended_widths"] = [len(t.matrix[0]) if t.matrix else 0 for t in stark.tables]
        rec["terminals"] = [limbs(t) for t in stark.get_terminals()]
        records.append(rec)
        print("%-50r %-8r cycles %4d  lengths %s" % (code[:48], inp[:6], running_time, rec["matrix_lengths"]), flush=True)
    out = {"python": sys.version.split()[0], "count": len(records), "programs": records}
    with open(os.path.join(HERE, "vm.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("wrote vm.json:", len(records), "programs")


if __name__ == "__main__":
    main()
