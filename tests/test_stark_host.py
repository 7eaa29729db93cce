"""Host-side logic of the STARK prover mirror against the goldens captured from the reference (tests/golden/stark_*.json,
made by tests/golden/gen_stark_golden.py): VM traces, FRI domain sizing, symbolic degree bounds, generated constraint code,
and the wire format (the reference's own proofs, read into this package's classes and written back, byte for byte).
No GPU: the native transcript code is host C++."""
import glob
import hashlib
import io
import json
import os
import random
import struct
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLDEN = os.path.join(HERE, "golden")
NAMES = sorted(os.path.basename(p)[len("stark_"):-len(".json")] for p in glob.glob(os.path.join(GOLDEN, "stark_*.json")))


def golden(name):
    return json.load(open(os.path.join(GOLDEN, "stark_%s.json" % name)))


def sha_rows(matrix):
    h = hashlib.sha256()
    for row in matrix:
        for e in row:
            h.update(struct.pack("<Q", e.value))
    return h.hexdigest()


@pytest.mark.parametrize("name", NAMES)
def test_vm_traces_and_domain_sizing(name):
    from stark_brainfuck_amd.brainfuck_stark import BrainfuckStark
    from stark_brainfuck_amd.vm import VirtualMachine
    g = golden(name)
    program = VirtualMachine.compile(g["program"])
    assert [w.value for w in program] == g["compiled_program"]
    running_time, inputs, outputs = VirtualMachine.run(program, input_data=list(g["input"]))
    assert running_time == g["running_time"] and "".join(outputs) == g["output"]
    matrices = VirtualMachine.simulate(program, input_data=list(inputs))
    for matrix, key in zip(matrices, ("processor", "memory", "instruction", "input", "output")):
        assert [len(matrix), len(matrix[0]) if matrix else g["matrix_shapes"][key][1]] == g["matrix_shapes"][key]
        assert sha_rows(matrix) == g["matrix_sha"][key], key
    stark = BrainfuckStark(running_time, len(matrices[1]), program, inputs, outputs)
    assert stark.max_degree == g["max_degree"]
    assert stark.fri.domain.length == g["fri_domain_length"]
    assert [t.height for t in stark.tables] == g["table_heights"]
    assert [t.unit_distance(stark.fri.domain.length) for t in stark.tables] == g["unit_distances"]
    assert stark.num_colinearity_checks == g["num_colinearity_checks"] and stark.expansion_factor == g["expansion_factor"]


@pytest.mark.parametrize("name", NAMES)
def test_quotient_degree_bounds(name):
    """exact symbolic expansion (air.expand) gives the reference's MPolynomial.symbolic_degree_bound for every quotient"""
    from stark_brainfuck_amd import air
    g = golden(name)
    randomizers = [1, 1, 1, 0, 0]
    for ti, (ta, q) in enumerate(zip(air.TABLE_AIRS, g["quotients"])):
        challenges = [tuple(c) for c in q["challenges"]]
        terminals = [tuple(t) for t in q["terminals"]]
        height, length = g["table_heights"][ti], g["table_lengths"][ti]
        md = height + randomizers[ti] - 1
        params = [air.xpow(challenges[ta.challenge_index], height - length)] if ta.num_params else []
        got = []
        for kind, constraints in ta.all():
            nvars = 2 * ta.full_width if kind == "transition" else ta.full_width
            for e in constraints:
                bound = air.symbolic_degree_bound(air.expand(e, nvars, challenges, terminals, params), md)
                got.append(bound - height + 1 if kind == "transition" else bound - 1)
        assert got == q["degree_bounds"], ta.name


def test_generated_constraint_code_matches_the_expression_graphs(tmp_path):
    """csrc/air_generated.hpp (what the quotient kernels run), compiled for the host, against air.evaluate at random
    points; also checks that the committed header is what tools/gen_air.py produces from air.py today."""
    from stark_brainfuck_amd import air
    header = os.path.join(ROOT, "stark_brainfuck_amd", "csrc", "air_generated.hpp")
    before = open(header).read()
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_air.py")], check=True, capture_output=True)
    assert open(header).read() == before, "air_generated.hpp is stale: run tools/gen_air.py"
    src = tmp_path / "chk.cpp"
    src.write_text(r'''
#include "%s"
#include <cstdio>
using namespace bfs;
int main() {
    u64 bc[16], bn[16]; Xfe xc[8], xn[8], ch[11], tm[5], pr[1]; unsigned long long v;
    auto rd = [&]() { if (scanf("%%llu", &v) != 1) return (u64)0; return (u64)v; };
    int table = (int)rd();
    for (int i = 0; i < 16; ++i) bc[i] = rd();
    for (int i = 0; i < 16; ++i) bn[i] = rd();
    for (int i = 0; i < 8; ++i) for (int l = 0; l < 3; ++l) xc[i].c[l] = rd();
    for (int i = 0; i < 8; ++i) for (int l = 0; l < 3; ++l) xn[i].c[l] = rd();
    for (int i = 0; i < 11; ++i) for (int l = 0; l < 3; ++l) ch[i].c[l] = rd();
    for (int i = 0; i < 5; ++i) for (int l = 0; l < 3; ++l) tm[i].c[l] = rd();
    for (int l = 0; l < 3; ++l) pr[0].c[l] = rd();
    Xfe out[32]; int n = 0;
    if (table == 0) { airgen::air_processor_values(bc, bn, xc, xn, ch, tm, pr, out); n = 21; }
    if (table == 1) { airgen::air_instruction_values(bc, bn, xc, xn, ch, tm, pr, out); n = 10; }
    if (table == 2) { airgen::air_memory_values(bc, bn, xc, xn, ch, tm, pr, out); n = 11; }
    if (table == 3) { airgen::air_input_values(bc, bn, xc, xn, ch, tm, pr, out); n = 3; }
    if (table == 4) { airgen::air_output_values(bc, bn, xc, xn, ch, tm, pr, out); n = 3; }
    for (int i = 0; i < n; ++i) printf("%%llu %%llu %%llu\n", (unsigned long long)out[i].c[0], (unsigned long long)out[i].c[1], (unsigned long long)out[i].c[2]);
}
''' % header)
    exe = tmp_path / "chk"
    subprocess.run(["g++", "-O1", "-std=c++17", "-o", str(exe), str(src)], check=True)
    rng = random.Random(11)
    P = air.P

    def rx():
        return tuple(rng.randrange(P) for _ in range(3))
    for ti, ta in enumerate(air.TABLE_AIRS):
        for trial in range(4):
            bc = [rng.randrange(P) for _ in range(16)]
            bn = [rng.randrange(P) for _ in range(16)]
            if trial < 2:
                bc[2] = ord(",.+-<>[]"[rng.randrange(8)])       # a real instruction in the current-instruction column
            xc, xn = [rx() for _ in range(8)], [rx() for _ in range(8)]
            ch, tm, pr = [rx() for _ in range(11)], [rx() for _ in range(5)], [rx()]
            flat = [ti] + bc + bn + [v for x in xc + xn + ch + tm + pr for v in x]
            out = subprocess.run([str(exe)], input=" ".join(map(str, flat)), capture_output=True, text=True, check=True).stdout.split()
            got = [tuple(int(out[3 * i + l]) for l in range(3)) for i in range(len(out) // 3)]
            bw = ta.base_width
            cur = [air.xlift(v) for v in bc[:bw]] + xc[:ta.full_width - bw]
            nxt = [air.xlift(v) for v in bn[:bw]] + xn[:ta.full_width - bw]
            want = [air.evaluate(e, cur, nxt, ch, tm, pr) for _, cons in ta.all() for e in cons]
            assert got == want, ta.name


@pytest.mark.parametrize("name", NAMES)
def test_reference_proofs_round_trip_byte_for_byte(name):
    """a proof written by the reference, read into this package's classes (identity of shared objects preserved by the
    unpickler) and serialised by the native transcript code, must come back identical: covers foreign BaseField
    instances, coefficient objects shared between elements, repeated rows, multi-object memoisation"""
    from stark_brainfuck_amd.ip import ProofStream
    path = os.path.join(GOLDEN, "stark_%s_proof.bin" % name)
    if not os.path.exists(path):
        pytest.skip("proof fixture not committed for this golden")
    proof = open(path, "rb").read()
    stream = ProofStream().deserialize(proof)
    assert len(stream.objects) == golden(name)["num_objects"]
    assert stream.serialize() == proof
    # the verifier's view after k objects is the prefix pickle: hash equal to the golden Fiat-Shamir seeds
    g = golden(name)
    stream.read_index = 1
    assert stream.verifier_fiat_shamir().hex() == g["fiat_shamir"][0]["seed"]
    stream.read_index = 7
    assert stream.verifier_fiat_shamir().hex() == g["fiat_shamir"][1]["seed"]


def test_running_evaluation_identity_rules():
    """processor_table._evaluation_step: which objects the reference's running evaluations are made of"""
    from stark_brainfuck_amd.algebra import BaseField, BaseFieldElement
    from stark_brainfuck_amd.processor_table import _evaluation_step
    f = BaseField.main()
    gamma = (5, 6, 7)
    a, b = BaseFieldElement(97, f), BaseFieldElement(98, f)
    state, ident = _evaluation_step((0, 0, 0), None, gamma, a)
    assert state == (97, 0, 0) and ident == ("object", a)            # zero + lift(a) IS a's polynomial
    state, ident = _evaluation_step(state, ident, gamma, b)
    assert ident == ("fresh", f) and state[1:] != (0, 0)             # afterwards new elements of a's field instance
    state, ident = _evaluation_step((0, 0, 0), None, gamma, BaseFieldElement(0, f))
    assert state == (0, 0, 0) and ident is None                      # lifting zero leaves no coefficient at all


@pytest.mark.parametrize("name", NAMES)
def test_evaluation_arguments_reproduce_the_terminals(name):
    """what the verifier recomputes from public data (evaluation_argument.py) equals the terminals the reference's prover sent"""
    from stark_brainfuck_amd.evaluation_argument import EvaluationArgument, ProgramEvaluationArgument
    from stark_brainfuck_amd.vm import VirtualMachine
    g = golden(name)
    program = VirtualMachine.compile(g["program"])
    _, inputs, outputs = VirtualMachine.run(program, input_data=list(g["input"]))
    challenges = [tuple(c) for c in g["quotients"][0]["challenges"]]
    assert list(ProgramEvaluationArgument([0, 1, 2, 10], 4, program).compute_terminal(challenges)) == g["terminals"][4]
    assert list(EvaluationArgument(8, 2, [ord(s) for s in inputs]).compute_terminal(challenges)) == g["terminals"][2]
    assert list(EvaluationArgument(9, 3, [ord(s) for s in outputs]).compute_terminal(challenges)) == g["terminals"][3]


def test_constraints_as_mpolynomials():
    """Table.*_constraints_ext give the reference's view of the AIR (lists of MPolynomial): evaluating them at a point must
    agree with the expression graphs, and their symbolic degree bounds with the golden ones"""
    import random
    from stark_brainfuck_amd import air
    from stark_brainfuck_amd.brainfuck_stark import BrainfuckStark
    from stark_brainfuck_amd.multivariate import MPolynomial
    from stark_brainfuck_amd.vm import VirtualMachine
    g = golden("io")
    program = VirtualMachine.compile(g["program"])
    running_time, inputs, outputs = VirtualMachine.run(program, input_data=list(g["input"]))
    matrices = VirtualMachine.simulate(program, input_data=list(inputs))
    stark = BrainfuckStark(running_time, len(matrices[1]), program, inputs, outputs)
    xf = stark.xfield
    rng = random.Random(3)
    for table, q in zip(stark.tables, g["quotients"]):
        challenges = [xf.from_limbs(c) for c in q["challenges"]]
        terminals = [xf.from_limbs(t) for t in q["terminals"]]
        md = [table.interpolant_degree()]
        polys = {"boundary": table.boundary_constraints_ext(challenges), "transition": table.transition_constraints_ext(challenges),
                 "terminal": table.terminal_constraints_ext(challenges, terminals)}
        bounds = ([p.symbolic_degree_bound(md * table.full_width) - 1 for p in polys["boundary"]]
                  + [p.symbolic_degree_bound(md * 2 * table.full_width) - table.height + 1 for p in polys["transition"]]
                  + [p.symbolic_degree_bound(md * table.full_width) - 1 for p in polys["terminal"]])
        assert bounds == q["degree_bounds"], type(table).__name__
        cur = [tuple(rng.randrange(air.P) for _ in range(3)) for _ in range(table.full_width)]
        nxt = [tuple(rng.randrange(air.P) for _ in range(3)) for _ in range(table.full_width)]
        ch, tm = [tuple(c) for c in q["challenges"]], [tuple(t) for t in q["terminals"]]
        for kind, ps in polys.items():
            point = [xf.from_limbs(v) for v in (cur + nxt if kind == "transition" else cur)]
            want = table.evaluate_constraints(kind, cur, nxt, ch, tm)
            for p, w in zip(ps, want):
                value = p.evaluate(point) if p.dictionary else xf.zero()
                assert tuple(value.limbs()) == tuple(w)
    x, y = MPolynomial.variables(2, xf)
    assert ((x + y) ^ 2).dictionary.keys() == {(2, 0), (1, 1), (0, 2)} and (x - x).is_zero()


@pytest.mark.parametrize("code,inp", [("++>+++<[->[->+>+<<]>>[-<<+>>]<<<]>>.", ""), (",>,<[->+<]>.", "!#"), (",.,.", "ab"), ("+", "")])
def test_extended_tables_satisfy_the_air(code, inp):
    """the restatement of Table.test / Table.xtest (/root/reference/code/table.py:48-110, used by test_vm.py:26-110): on the
    padded and extended trace every boundary constraint vanishes on the first row, every transition constraint on every pair
    of consecutive rows, every terminal constraint on the last row -- ties VM, padding, the native scans and air.py together"""
    import numpy as np
    from stark_brainfuck_amd import air
    from stark_brainfuck_amd.brainfuck_stark import BrainfuckStark
    from stark_brainfuck_amd.vm import VirtualMachine
    program = VirtualMachine.compile(code)
    running_time, inputs, outputs = VirtualMachine.run(program, input_data=list(inp))
    matrices = VirtualMachine.simulate(program, input_data=list(inputs))
    stark = BrainfuckStark(running_time, len(matrices[1]), program, inputs, outputs)
    rng = np.random.default_rng(1)
    challenges = [tuple(int(v) for v in rng.integers(1, air.P, 3, dtype=np.uint64)) for _ in range(11)]
    initials = [tuple(int(v) for v in rng.integers(1, air.P, 3, dtype=np.uint64)) for _ in range(2)]
    for table, matrix in zip(stark.tables, (matrices[0], matrices[2], matrices[1], matrices[3], matrices[4])):
        table.matrix = matrix
        table.pad()
        table.extend(challenges, initials)
    terminals = stark.get_terminals()
    # the permutation arguments: processor and instruction / memory tables reach the same terminal (running products over
    # the same multiset), which is what lets the difference quotients be polynomials
    assert stark.processor_table.instruction_permutation_terminal == stark.instruction_table.permutation_terminal
    assert stark.processor_table.memory_permutation_terminal == stark.memory_table.permutation_terminal
    for table in stark.tables:
        base = table.base_array()
        height = base.shape[1]
        if height == 0:
            continue
        rows = []
        for r in range(height):
            row = [air.xlift(int(base[c, r])) for c in range(table.base_width)]
            row += [tuple(int(v) for v in col[:, r]) for col in table.ext_columns]
            rows.append(row)
        for value in table.evaluate_constraints("boundary", rows[0], None, challenges, terminals):
            assert value == air.X0, type(table).__name__
        for r in range(height - 1):
            for k, value in enumerate(table.evaluate_constraints("transition", rows[r], rows[r + 1], challenges, terminals)):
                assert value == air.X0, (type(table).__name__, r, k)
        for value in table.evaluate_constraints("terminal", rows[-1], None, challenges, terminals):
            assert value == air.X0, type(table).__name__


def test_generic_degree_bound_cache_agrees_with_the_exact_expansion():
    """Table._degree_bounds reuses the surviving-monomial pattern for sampled-looking challenges; the result must equal the
    exact expansion with those very values, for zero and non-zero terminals, and crafted small values must bypass the cache"""
    import numpy as np
    from stark_brainfuck_amd import air
    from stark_brainfuck_amd.brainfuck_stark import BrainfuckStark
    from stark_brainfuck_amd.table import Table
    from stark_brainfuck_amd.vm import VirtualMachine
    program = VirtualMachine.compile(",+.")
    running_time, inputs, outputs = VirtualMachine.run(program, input_data=["a"])
    matrices = VirtualMachine.simulate(program, input_data=list(inputs))
    stark = BrainfuckStark(running_time, len(matrices[1]), program, inputs, outputs)
    rng = np.random.default_rng(7)
    sample = lambda k: [tuple(int(v) for v in rng.integers(1 << 40, air.P, 3, dtype=np.uint64)) for _ in range(k)]
    for trial in range(3):
        challenges = sample(11)
        terminals = sample(5)
        if trial == 1:
            terminals[2] = terminals[3] = air.X0          # a program without input / output
        for table in stark.tables:
            for kind in ("boundary", "transition", "terminal"):
                exact = [max([-1] + [t * table.interpolant_degree() for t in ts])
                         for ts in table._constraint_total_degrees(kind, challenges, terminals, table.air_params(challenges))]
                assert table._degree_bounds(kind, challenges, terminals) == exact, (type(table).__name__, kind)
    cached = len(Table._generic_totals)
    small = [(i + 2, 0, 0) for i in range(11)]
    for table in stark.tables:
        table._degree_bounds("transition", small, [air.X0] * 5)
    assert len(Table._generic_totals) == cached             # crafted values: exact path, nothing new cached


def test_threaded_extension_equals_the_sequential_one(monkeypatch):
    """table.extend_all (worker threads, used for long traces) gives the columns and terminals of plain Table.extend calls"""
    import numpy as np
    from stark_brainfuck_amd import air, table as table_module
    from stark_brainfuck_amd.brainfuck_stark import BrainfuckStark
    from stark_brainfuck_amd.vm import VirtualMachine
    program = VirtualMachine.compile("++>+++<[->[->+>+<<]>>[-<<+>>]<<<]>>.")
    running_time, inputs, outputs = VirtualMachine.run(program)
    matrices = VirtualMachine.simulate(program, input_data=list(inputs))
    rng = np.random.default_rng(3)
    challenges = [tuple(int(v) for v in rng.integers(1, air.P, 3, dtype=np.uint64)) for _ in range(11)]
    initials = [tuple(int(v) for v in rng.integers(1, air.P, 3, dtype=np.uint64)) for _ in range(2)]
    results = []
    for threaded in (False, True):
        stark = BrainfuckStark(running_time, len(matrices[1]), program, inputs, outputs)
        for t, matrix in zip(stark.tables, (matrices[0], matrices[2], matrices[1], matrices[3], matrices[4])):
            t.matrix = matrix
            t.pad()
        if threaded:
            monkeypatch.setattr(table_module, "_THREAD_ROWS", 1)
            table_module.extend_all(stark.tables, challenges, initials)
        else:
            for t in stark.tables:
                t.extend(challenges, initials)
        results.append(([np.concatenate(t.ext_columns).tobytes() if t.height else b"" for t in stark.tables], stark.get_terminals()))
    assert results[0] == results[1]


def _random_program(rng, length):
    """-> (code, input string) of a random well-bracketed Brainfuck program that terminates quickly.  The generator runs the
    machine while it writes the program: `-` is only emitted on a cell that stays >= 0 and `<` only right of cell 0, so every loop
    ([-] and [->+<]) starts on a small non-negative cell.  (Decrementing below zero wraps to p - 1 and a loop on such a cell
    runs ~1.8e19 iterations; that case is covered by the cycle-limit test, not here.)"""
    out, inputs = [], []
    cells, mp = {}, 0
    while len(out) < length:
        k = rng.integers(0, 10)
        if k < 4:
            count = int(rng.integers(1, 4))
            if rng.integers(0, 2) and cells.get(mp, 0) >= count:
                out.append("-" * count)
                cells[mp] = cells.get(mp, 0) - count
            else:
                out.append("+" * count)
                cells[mp] = cells.get(mp, 0) + count
        elif k < 6:
            if rng.integers(0, 2) and mp > 0:
                out.append("<")
                mp -= 1
            else:
                out.append(">")
                mp += 1
        elif k == 6:
            out.append(",")
            inputs.append(int(rng.integers(1, 120)))
            cells[mp] = inputs[-1]
        elif k == 7:
            out.append(".")
        elif k == 8:
            out.append("[-]")
            cells[mp] = 0
        else:
            out.append("[->+<]")
            cells[mp + 1] = cells.get(mp + 1, 0) + cells.get(mp, 0)
            cells[mp] = 0
    return "".join(out), "".join(chr(c) for c in inputs)


def test_native_vm_equals_the_object_building_one():
    """bfs_vm_trace_new (csrc/vm.cpp) against the element-by-element restatement of vm.py:172-306 on the golden programs and on random
    ones: the five matrices, and which entries of the memory-value column / input / output matrices are the SAME object"""
    import numpy as np
    from stark_brainfuck_amd.vm import VirtualMachine
    rng = np.random.default_rng(11)
    cases = [(golden(name)["program"], golden(name)["input"]) for name in NAMES]
    for _ in range(25):
        cases.append(_random_program(rng, int(rng.integers(1, 40))))
    cases.append(("-<-.", ""))                       # wraps the memory pointer and the value below zero (mod p)
    for code, inp in cases:
        program = VirtualMachine.compile(code)
        native = VirtualMachine.simulate(program, input_data=list(inp))
        objects = VirtualMachine.simulate_objects(program, input_data=list(inp))
        for a, b, name in zip(native, objects, ("processor", "memory", "instruction", "input", "output")):
            assert len(a) == len(b), (code, name)
            assert [[e.value for e in row] for row in a] == [[e.value for e in row] for row in b], (code, name)
            assert (a.values == b.values).all()
        # identity classes of the memory-value objects, numbered by first appearance
        def classes(matrices):
            seen, out = {}, []
            for m, col in ((matrices[0], 5), (matrices[3], 0), (matrices[4], 0)):
                out.append([seen.setdefault(id(row[col]), len(seen)) for row in m])
            return out
        assert classes(native) == classes(objects), code
        assert all(row[5].field is VirtualMachine.field for row in native[0])
    # the reference's error behaviour
    with pytest.raises(AssertionError, match="more input symbols"):
        VirtualMachine.simulate(VirtualMachine.compile(",,"), input_data=["a"])


def test_runaway_programs_stop_at_the_cycle_limit():
    """`-[-]` counts down from p - 1 (~1.8e19 iterations): every entry point that runs a program gives up with an AssertionError
    at its cycle limit -- quickly, and without growing a trace until the process is killed (not in the reference, whose
    vm.py:107-165 / 172-306 would simply never return)"""
    import time
    from stark_brainfuck_amd.vm import VirtualMachine
    program = VirtualMachine.compile("-[-]")
    t0 = time.perf_counter()
    with pytest.raises(AssertionError, match="more than 100000 cycles"):
        VirtualMachine.simulate(program, max_cycles=100000)
    with pytest.raises(AssertionError, match="more than 100000 cycles"):
        VirtualMachine.run(program, max_cycles=100000)
    with pytest.raises(AssertionError, match="more than 5000 cycles"):
        VirtualMachine.simulate_objects(program, max_cycles=5000)
    assert time.perf_counter() - t0 < 120.0        # ~1.4 s here (18 s seen once on a cold page cache); the bound only has to tell "stops" from "runs away"
    # the default limit (no argument) is finite as well: 2^24 cycles of the native machine
    assert VirtualMachine.DEFAULT_MAX_CYCLES == 1 << 24
    t0 = time.perf_counter()
    with pytest.raises(AssertionError, match=f"more than {1 << 24} cycles"):
        VirtualMachine.simulate(program)
    assert time.perf_counter() - t0 < 300.0        # ~2 s here
    # a program that stays under the limit is not affected by it
    assert len(VirtualMachine.simulate(VirtualMachine.compile("+++[-]"), max_cycles=100)[0]) == 11
    # 0 and None mean the default (as in bfs_vm_trace_new), UNLIMITED means the reference's behaviour (no cap)
    assert VirtualMachine._cycle_limit(0) == VirtualMachine._cycle_limit(None) == 1 << 24
    assert VirtualMachine._cycle_limit(VirtualMachine.UNLIMITED) == (1 << 64) - 1
    short = VirtualMachine.compile("+++[-]")
    assert len(VirtualMachine.simulate(short, max_cycles=0)[0]) == len(VirtualMachine.simulate(short, max_cycles=VirtualMachine.UNLIMITED)[0]) == 11
    assert VirtualMachine.run(short, max_cycles=0)[0] == VirtualMachine.run(short, max_cycles=VirtualMachine.UNLIMITED)[0]


def _vm_golden():
    return json.load(open(os.path.join(GOLDEN, "vm.json")))["programs"]


def test_vm_pad_extend_against_the_reference_on_many_programs():
    """tests/golden/vm.json (gen_vm_golden.py: the reference's VirtualMachine, Table.pad and Table.extend on 91 programs -- loops skipped
    on a zero cell, cells and the memory pointer wrapping below zero, nested loops, input / output inside loops, empty tables, random
    programs): compiled words, running time, output, the five trace matrices, the five padded tables, the five extended tables under
    the fixture's challenges and initials, and the terminals.  vm.py:172-306, processor_table.py:24-31 / 359-427 and the other tables."""
    import numpy as np
    from stark_brainfuck_amd.brainfuck_stark import BrainfuckStark
    from stark_brainfuck_amd.vm import VirtualMachine
    records = _vm_golden()
    assert len(records) >= 90
    checked = errors = 0
    for g in records:
        code, inp = g["program"], g["input"]
        program = VirtualMachine.compile(code)
        assert [w.value for w in program] == g["compiled_program"], code
        if "run_error" in g:
            # the reference's run() raises KeyError when `.` reads a cell nothing was ever written to (vm.py:148); so does the mirror
            with pytest.raises(KeyError):
                VirtualMachine.run(program, input_data=list(inp))
            errors += 1
            continue
        running_time, inputs, outputs = VirtualMachine.run(program, input_data=list(inp))
        assert running_time == g["running_time"] and "".join(outputs) == g["output"] and "".join(inputs) == g["input_symbols"], code
        matrices = VirtualMachine.simulate(program, input_data=list(inputs))
        assert [len(m) for m in matrices] == g["matrix_lengths"], code
        assert [sha_rows(m) for m in matrices] == g["matrix_sha"], code
        stark = BrainfuckStark(running_time, len(matrices[1]), program, inputs, outputs)
        assert stark.max_degree == g["max_degree"] and stark.fri.domain.length == g["fri_domain_length"], code
        assert [t.height for t in stark.tables] == g["table_heights"], code
        for table, matrix in zip(stark.tables, (matrices[0], matrices[2], matrices[1], matrices[3], matrices[4])):
            table.matrix = matrix
            table.pad()
        padded = []
        for table in stark.tables:
            base = table.base_array()                                        # width x height
            padded.append(hashlib.sha256(np.ascontiguousarray(base.T).astype("<u8").tobytes()).hexdigest())
            assert base.shape[1] == g["padded_lengths"][len(padded) - 1], code
        assert padded == g["padded_sha"], code
        challenges = [tuple(c) for c in g["challenges"]]
        initials = [tuple(c) for c in g["initials"]]
        for table in stark.tables:
            table.extend(challenges, initials)
        for k, table in enumerate(stark.tables):
            base = table.base_array()
            height = base.shape[1]
            width = table.base_width + len(table.ext_columns)
            assert (width if height else 0) == g["extended_widths"][k], (code, k)
            rows = np.zeros((height, width, 3), dtype="<u8")
            rows[:, :table.base_width, 0] = base.T
            for j, col in enumerate(table.ext_columns):
                rows[:, table.base_width + j, :] = np.asarray(col, dtype=np.uint64).T
            assert hashlib.sha256(rows.tobytes()).hexdigest() == g["extended_sha"][k], (code, type(table).__name__)
        assert [list(t) for t in stark.get_terminals()] == g["terminals"], code
        checked += 1
    assert checked >= 75 and errors >= 5


def test_table_caches_follow_the_matrix_object_not_its_id():
    """Table caches the column-major copy of its matrix and the scan masks derived from it.  The same table objects are given one
    trace after another of the SAME height (prove() called again, tools, tests); the caches must follow the matrix object -- the old key,
    (id(matrix), rows), comes back when CPython reuses a freed matrix' address (round-3 advice)"""
    import gc
    import numpy as np
    from stark_brainfuck_amd.brainfuck_stark import BrainfuckStark
    from stark_brainfuck_amd.vm import VirtualMachine
    programs = [VirtualMachine.compile(c) for c in ("++>+++.<-.", "+>++<+.>-.")]         # equal lengths and running times, different traces
    runs = [VirtualMachine.run(p) for p in programs]
    assert runs[0][0] == runs[1][0]

    def masks_of(stark, matrices):
        order = (matrices[0], matrices[2], matrices[1], matrices[3], matrices[4])
        for table, matrix in zip(stark.tables, order):
            table.matrix = matrix
        for table in stark.tables:
            table.pad()
        return [[None if m is None else np.array(m, copy=True) for m in t._scan_masks()] for t in stark.tables], [t.base_array().copy() for t in stark.tables]

    def same(a, b):
        return all((x is None and y is None) or (x is not None and y is not None and np.array_equal(x, y)) for ta, tb in zip(a, b) for x, y in zip(ta, tb))
    fresh = []
    for program, (rt, inputs, outputs) in zip(programs, runs):
        stark = BrainfuckStark(rt, 0, program, inputs, outputs)
        fresh.append(masks_of(stark, VirtualMachine.simulate(program, input_data=inputs)))
    assert not same(fresh[0][0], fresh[1][0]) or not all(np.array_equal(a, b) for a, b in zip(fresh[0][1], fresh[1][1]))
    rt, inputs, outputs = runs[0]
    reused = BrainfuckStark(rt, 0, programs[0], inputs, outputs)
    for rep in range(6):
        k = rep % 2
        matrices = VirtualMachine.simulate(programs[k], input_data=runs[k][1])
        masks, arrays = masks_of(reused, matrices)
        assert same(masks, fresh[k][0]), rep
        assert all(np.array_equal(a, b) for a, b in zip(arrays, fresh[k][1])), rep
        del matrices, masks, arrays
        for t in reused.tables:
            t.matrix = []                      # drop the padded matrices so that their addresses are free for the next round
        gc.collect()


def test_native_reader_of_proof_streams_matches_the_python_route():
    """bfs_ps_loads (round 4: the verifier's proof stream is read natively instead of walking the unpickled Python objects): for every
    proof and FRI transcript the reference wrote, the native stream exists (i.e. it serialises back to the input byte for byte), the
    Fiat-Shamir bytes over EVERY prefix equal those of the stream built from Python objects, and the pickle of every top-level object
    (and of the items of top-level tuples: FRI's leaves) equals the Python route's -- a Merkle leaf preimage is exactly that."""
    import glob
    import pickle
    from stark_brainfuck_amd.ip import NativeTranscript, ProofStream, reference_pickle
    files = sorted(glob.glob(os.path.join(GOLDEN, "stark_*_proof.bin")) + glob.glob(os.path.join(GOLDEN, "fri_*_stream.bin")))
    assert len(files) >= 14
    for path in files:
        data = open(path, "rb").read()
        ps = ProofStream().deserialize(data)
        native = getattr(ps, "_cached", None)
        assert native is not None and native.loaded, path
        assert ps.serialize() == data
        slow = ProofStream()
        slow.objects = list(ps.objects)               # the same Python objects through the Python -> native walk
        assert not slow._native().loaded
        step = max(1, len(ps.objects) // 40)
        for k in list(range(0, len(ps.objects) + 1, step)) + [len(ps.objects)]:
            ps.read_index = slow.read_index = k
            assert ps.verifier_fiat_shamir() == slow.verifier_fiat_shamir(), (path, k)
        for o in ps.objects[::7]:
            assert ps.pickle_of(o) == reference_pickle(o), path
            if isinstance(o, tuple):
                for c in o:
                    assert ps.pickle_of(c) == reference_pickle(c), path
        assert ps.pickle_of(b"not in the stream") is None
        # appending to a natively read stream falls back to the walk (and still gives the reference's bytes)
        more = ProofStream().deserialize(data)
        more.push(more.objects[0])
        expect = ProofStream()
        expect.objects = list(more.objects)
        assert more.serialize() == expect.serialize()
    # what the native reader does not take is still read, through CPython's unpickler alone
    odd = pickle.dumps([b"\x01" * 64, None, True, -5], protocol=4)
    assert NativeTranscript.from_bytes(odd) is None
    ps = ProofStream().deserialize(odd)
    assert ps.objects == [b"\x01" * 64, None, True, -5] and getattr(ps, "_cached", None) is None
    assert NativeTranscript.from_bytes(data[:-7]) is None and NativeTranscript.from_bytes(b"") is None
    # bytes from a hostile prover: a list nested 100 000 deep must be refused by the native reader (its pickler is recursive), not crash it
    for depth in (1000, 100000, 3000000):
        deep = b"\x80\x04" + b"]" * depth + b"a" * (depth - 1) + b"."
        assert NativeTranscript.from_bytes(deep) is None
    # round-4 advice: nesting does not need stack depth -- TUPLE1 (0x85) pops one object and pushes one, so two million of them build a
    # chain two million deep with one object on the stack; freeing such a graph recursively overflowed the C stack (rc 139).  Run in a
    # child process so that a crash fails this test instead of ending the suite.
    import subprocess
    import sys
    hostile = r"""
import sys
sys.path.insert(0, %r)
from stark_brainfuck_amd.ip import NativeTranscript, ProofStream
chain = b'\x80\x04])' + b'\x85' * 2000000 + b'a.'
assert NativeTranscript.from_bytes(chain) is None
from stark_brainfuck_amd import _lib
assert b'deeper than 200' in _lib.load().bfs_last_error()
ps = ProofStream().deserialize(chain)                       # CPython reads it: the Python route ends in an ordinary object
assert isinstance(ps.objects, list) and len(ps.objects) == 1
# depth hidden from bookkeeping done at attach time: A_k+1 = [B_k+1] is two levels deep when it goes into B_k, and B_k already
# sits inside A_k, so no node is ever seen deeper than 3 while it is attached; the finished graph is 800 levels deep from A_0 on
def get(i):
    return b'j' + i.to_bytes(4, 'little')
parts = [b'\x80\x04]\x94(']                                # the outer list (memo 0), MARK
memo, prev_b = 1, None
for k in range(400):
    parts.append(b']\x94]\x94a')                            # A_k = [B_k]; memo: A_k, B_k
    a, b = memo, memo + 1
    memo += 2
    if prev_b is not None:
        parts.append(get(prev_b) + get(a) + b'a')             # B_k-1.append(A_k)
    prev_b = b
parts.append(b'e.')                                           # outer.extend(everything on the stack)
hidden = b''.join(parts)
import pickle
assert len(pickle.loads(hidden)) == 799                       # a pickle CPython reads
assert NativeTranscript.from_bytes(hidden) is None
from stark_brainfuck_amd import _lib
assert b'deeper than 200' in _lib.load().bfs_last_error()
# a list that contains itself, and a two-node cycle through a tuple
assert NativeTranscript.from_bytes(b'\x80\x04]\x94h\x00a.') is None
assert NativeTranscript.from_bytes(b'\x80\x04]\x94]\x94h\x00\x85ah\x01a.') is None
print('survived')
""" % ROOT
    res = subprocess.run([sys.executable, "-c", hostile], capture_output=True, text=True)
    assert res.returncode == 0 and "survived" in res.stdout, (res.returncode, res.stderr[-2000:])
    # a pickle that CPython would lay out differently (protocol 2 of the same list) is refused by the round-trip check, not misread
    assert NativeTranscript.from_bytes(pickle.dumps([b"ab", 7], protocol=2)) is None


def test_native_constraint_evaluation_matches_the_expression_graphs():
    """bfs_air_evaluate (the generated constraint code compiled for the host; what verify() calls since round 4) against air.evaluate
    walking the expression graphs in Python, for every table, on random rows, challenges, terminals and parameters -- including base
    values at the edges of the field and all-zero extension elements"""
    import random
    from stark_brainfuck_amd.brainfuck_stark import BrainfuckStark
    from stark_brainfuck_amd.vm import VirtualMachine
    P = (1 << 64) - (1 << 32) + 1
    program = VirtualMachine.compile("++>,<[>+.<-]")
    running_time, inputs, outputs = VirtualMachine.run(program, input_data=["a"])
    stark = BrainfuckStark(running_time, 8, program, inputs, outputs)
    rnd = random.Random(20260930)
    edge = [0, 1, P - 1, P - 2, (1 << 32) - 1, 1 << 32, (1 << 63) + 5]

    def felt():
        return rnd.choice(edge) if rnd.random() < 0.3 else rnd.randrange(P)

    def xfelt():
        return (0, 0, 0) if rnd.random() < 0.1 else (felt(), felt(), felt())
    for trial in range(40):
        challenges = tuple(xfelt() for _ in range(11))
        terminals = [xfelt() for _ in range(5)]
        for table in stark.tables:
            bw, xw = table.base_width, table.full_width - table.base_width
            point = [(felt(), 0, 0) for _ in range(bw)] + [xfelt() for _ in range(xw)]
            nxt = [(felt(), 0, 0) for _ in range(bw)] + [xfelt() for _ in range(xw)]
            got = table.evaluate_all_constraints(point, nxt, challenges, terminals)
            for kind, values in zip(("boundary", "transition", "terminal"), got):
                want = table.evaluate_constraints(kind, point, nxt, challenges, terminals)
                assert [tuple(v) for v in want] == [tuple(v) for v in values], (trial, type(table).__name__, kind)
            # round-4 advice: the same elements in another representation (v + p where that fits 64 bits -- what a prover may pickle
            # into a proof) must evaluate to the same values: bfs_air_evaluate reduces every operand on the way in
            def other(v):
                return v + P if v + P < (1 << 64) and rnd.random() < 0.7 else v

            def recode(row):
                return [tuple(other(v) for v in e) for e in row]
            again = table.evaluate_all_constraints(recode(point), recode(nxt), tuple(recode(challenges)), recode(terminals))
            assert [[tuple(v) for v in part] for part in again] == [[tuple(v) for v in part] for part in got], (trial, type(table).__name__)


@pytest.mark.parametrize("name", NAMES)
def test_verify_on_the_host_accepts_reference_proofs_and_rejects_tampering(name):
    """the verifier mirror (brainfuck_stark.py:343-579) on proofs written by the reference itself, WITHOUT a GPU: since round 4 the
    verifier's one tree (the last FRI codeword) is hashed with hashlib like every path check, so verify() is host code throughout, as in
    the reference"""
    from stark_brainfuck_amd.brainfuck_stark import BrainfuckStark
    from stark_brainfuck_amd.vm import VirtualMachine
    g = json.load(open(os.path.join(GOLDEN, "stark_%s.json" % name)))
    path = os.path.join(GOLDEN, "stark_%s_proof.bin" % name)
    if not os.path.exists(path):
        pytest.skip("no proof bytes committed for this fixture")
    program = VirtualMachine.compile(g["program"])
    running_time, input_symbols, output_symbols = VirtualMachine.run(program, input_data=list(g["input"]))
    _, mm, _, _, _ = VirtualMachine.simulate(program, input_data=list(input_symbols))
    proof = open(path, "rb").read()
    stark = BrainfuckStark(running_time, len(mm), program, input_symbols, output_symbols)
    assert stark.verify(proof) is True
    other = BrainfuckStark(running_time, len(mm), program, input_symbols, list(output_symbols) + ["!"])
    try:
        assert other.verify(proof) is False          # a claim about a different output
    except AssertionError:
        pass
    pos = proof.index(bytes.fromhex(g["combination_tree"]["root"])) + 200
    bad = bytearray(proof)
    bad[pos] ^= 1                                     # one bit inside an opened digest
    try:
        assert stark.verify(bytes(bad)) is False
    except (AssertionError, Exception):
        pass


def test_an_instance_that_was_never_built_is_refused_by_the_native_reader():
    """found by tests/test_sanitized_parsers.py in round 6: NEWOBJ followed by neither MEMOIZE nor BUILD leaves an instance without state
    in the graph; only memoised instances were checked at STOP, and the round-trip check of bfs_ps_loads dereferenced the missing state
    (SIGSEGV on two byte mutations of a reference proof).  The reader refuses such a stream now -- the synthetic one below and the fuzzer's
    own input (tests/golden/hostile_unbuilt_instance.bin, a mutated reference proof) -- and verify() falls back to the Python route, which
    rejects it like any other malformed proof."""
    from stark_brainfuck_amd import _lib
    from stark_brainfuck_amd.ip import NativeTranscript
    tiny = b"\x80\x04]\x8c\x07algebra\x8c\tBaseField\x93)\x81a."
    assert NativeTranscript.from_bytes(tiny) is None
    assert b"an instance without state" in _lib.load().bfs_last_error()
    hostile = open(os.path.join(GOLDEN, "hostile_unbuilt_instance.bin"), "rb").read()
    assert NativeTranscript.from_bytes(hostile) is None
    assert b"an instance without state" in _lib.load().bfs_last_error()
    from tools.fuzz_proofs import stark_of
    for name in ("mul34", "two_io"):
        try:
            assert stark_of(name).verify(hostile) is not True
        except Exception:       # noqa: BLE001 -- the reference's verifier raises on malformed streams too
            pass


@pytest.mark.parametrize("native", ["1", "0"])
def test_terminal_stored_as_v_plus_p_gets_the_reference_verdict(native, monkeypatch):
    """round-5 advice.  tests/golden/noncanonical_terminal_proof.bin was written by the REFERENCE's prover (gen_noncanonical_terminal.py)
    with the zero input-evaluation terminal pushed as the one-coefficient polynomial [p]: Fiat-Shamir, Merkle paths, the AIR, FRI -- all
    consistent, every arithmetic use of the terminal reduces -- and the reference's verify says False, because `ea.select_terminal(terminals)
    == ea.compute_terminal(challenges)` (brainfuck_stark.py:574-577) compares stored coefficient values (univariate.py:67-74, algebra.py:
    48-49).  Both routes here must say False as well (they said True while they compared reduced limbs); the honest proof of the same
    claim, made in the same run of the reference, is accepted."""
    from stark_brainfuck_amd.brainfuck_stark import BrainfuckStark
    from stark_brainfuck_amd.vm import VirtualMachine
    path = os.path.join(GOLDEN, "noncanonical_terminal.json")
    if not os.path.exists(path):
        pytest.skip("fixture not generated")
    g = json.load(open(path))
    assert g["honest"]["reference_verify"] is True and g["crafted"]["reference_verify"] is False
    monkeypatch.setenv("BFS_NATIVE_VERIFY", native)
    program = VirtualMachine.compile(g["program"])
    running_time, input_symbols, output_symbols = VirtualMachine.run(program, input_data=list(g["input"]))
    _, mm, _, _, _ = VirtualMachine.simulate(program, input_data=list(input_symbols))
    args = (running_time, len(mm), program, input_symbols, output_symbols)
    crafted = open(os.path.join(GOLDEN, "noncanonical_terminal_proof.bin"), "rb").read()
    honest = open(os.path.join(GOLDEN, "noncanonical_terminal_honest_proof.bin"), "rb").read()
    assert hashlib.sha256(crafted).hexdigest() == g["crafted"]["proof_sha256"]
    assert BrainfuckStark(*args).verify(honest) is True
    assert BrainfuckStark(*args).verify(crafted) is False


def _verdict(stark_args, proof, native):
    """('value', bool) / ('assert', message) / ('error', exception type) of BrainfuckStark(*stark_args).verify(proof) on one of the two routes"""
    from stark_brainfuck_amd.brainfuck_stark import BrainfuckStark
    old = os.environ.get("BFS_NATIVE_VERIFY")
    os.environ["BFS_NATIVE_VERIFY"] = "1" if native else "0"
    try:
        return ("value", BrainfuckStark(*stark_args).verify(proof))
    except AssertionError as e:
        return ("assert", str(e))
    except Exception as e:          # noqa: BLE001 -- whatever the Python verifier raises, the native route must end in the same
        return ("error", type(e).__name__)
    finally:
        if old is None:
            os.environ.pop("BFS_NATIVE_VERIFY", None)
        else:
            os.environ["BFS_NATIVE_VERIFY"] = old


@pytest.mark.parametrize("name", ["plus1", "io", "two_io", "loop", "countdown"])
def test_native_verifier_agrees_with_the_python_verifier_on_mutated_proofs(name):
    """csrc/verifier.cpp (bfs_stark_verify_begin / _finish: the verifier on the native object graph of the proof, round 5) against
    _verify_stream / Fri.verify in Python, which mirror brainfuck_stark.py:343-579 and fri.py:201-319: the reference's proof and ~70 mutations of
    it at the OBJECT level -- a bit in a root, a limb of an opened element (also as the non-canonical v + p), an element replaced by one of the
    other kind, a salt, a path node, a path of the wrong length, a leaf of a FRI triple, an element of the last codeword, objects dropped /
    doubled / swapped, a truncated stream, a different claim -- must end the same way on both routes: the same boolean, the same AssertionError
    message, or the same exception type (the native route hands anything it does not model to the Python verifier)."""
    import random
    from stark_brainfuck_amd import ExtensionFieldElement, BaseFieldElement, ProofStream
    from stark_brainfuck_amd.vm import VirtualMachine
    g = json.load(open(os.path.join(GOLDEN, "stark_%s.json" % name)))
    proof = open(os.path.join(GOLDEN, "stark_%s_proof.bin" % name), "rb").read()
    program = VirtualMachine.compile(g["program"])
    running_time, input_symbols, output_symbols = VirtualMachine.run(program, input_data=list(g["input"]))
    _, mm, _, _, _ = VirtualMachine.simulate(program, input_data=list(input_symbols))
    args = (running_time, len(mm), program, input_symbols, output_symbols)
    assert _verdict(args, proof, True) == _verdict(args, proof, False) == ("value", True)
    wrong = (running_time, len(mm), program, input_symbols, list(output_symbols) + ["!"])
    assert _verdict(wrong, proof, True) == _verdict(wrong, proof, False)
    P = (1 << 64) - (1 << 32) + 1
    rnd = random.Random(hash(name) & 0xFFFF)
    base = ProofStream().deserialize(proof).objects
    xf = next(o for o in base if isinstance(o, ExtensionFieldElement)).field

    def flip(b):
        k = rnd.randrange(len(b))
        return b[:k] + bytes([b[k] ^ (1 << rnd.randrange(8))]) + b[k + 1:]

    def bump(e):
        """another element in place of e: a limb changed, the same value as v + p, or an element of the other kind"""
        if isinstance(e, ExtensionFieldElement):
            limbs = list(e.limbs())
            how = rnd.randrange(4)
            if how == 0:
                limbs[rnd.randrange(3)] = (limbs[rnd.randrange(3)] + 1) % P
            elif how == 1:
                k = rnd.randrange(3)
                if limbs[k] + P < (1 << 64):
                    limbs[k] += P                           # the same field element in another representation
                else:
                    limbs[k] = (limbs[k] + 5) % P
            elif how == 2:
                return BaseFieldElement(limbs[0], xf.modulus.coefficients[0].field)
            else:
                limbs = [rnd.randrange(P) for _ in range(3)]
            return xf.from_limbs(limbs)
        if isinstance(e, BaseFieldElement):
            return BaseFieldElement((e.value + 1) % P, e.field) if rnd.random() < 0.7 else xf.from_limbs([e.value, 0, 1])
        return e

    def mutate(o):
        if isinstance(o, (bytes, bytearray)):
            return flip(bytes(o))
        if isinstance(o, (ExtensionFieldElement, BaseFieldElement)):
            return bump(o)
        if isinstance(o, tuple) and len(o) == 2 and isinstance(o[0], (bytes, bytearray)):       # (salt, path)
            if rnd.random() < 0.4:
                return (flip(o[0]), o[1])
            return (o[0], mutate(o[1]))
        if isinstance(o, (list, tuple)) and len(o):
            items = list(o)
            how = rnd.randrange(5)
            if how == 0 and len(items) > 1:
                del items[rnd.randrange(len(items))]
            elif how == 1:
                items.append(items[-1])
            else:
                k = rnd.randrange(len(items))
                items[k] = mutate(items[k])
            return type(o)(items)
        return o

    tried = 0
    for trial in range(70):
        objs = list(base)
        how = rnd.randrange(10)
        if how == 0:
            del objs[rnd.randrange(len(objs))]
        elif how == 1:
            k = rnd.randrange(len(objs))
            objs.insert(k, objs[k])
        elif how == 2:
            i, j = rnd.randrange(len(objs)), rnd.randrange(len(objs))
            objs[i], objs[j] = objs[j], objs[i]
        elif how == 3:
            objs = objs[:rnd.randrange(1, len(objs))]
        else:
            k = rnd.randrange(min(len(objs), 8)) if rnd.random() < 0.3 else rnd.randrange(len(objs))
            objs[k] = mutate(objs[k])
        ps = ProofStream()
        ps.objects = objs
        try:
            data = ps.serialize()
        except TypeError:
            continue
        tried += 1
        a, b = _verdict(args, data, True), _verdict(args, data, False)
        assert a == b, (trial, how, a, b)
    assert tried >= 50


@pytest.mark.parametrize("tag", ["d16_t2", "d64_t8", "d1024_t4", "test_fri_valid", "test_fri_disturbed", "d16_t2_prepushed"])
def test_fri_verify_on_the_host_golden_transcripts(tag):
    """Fri.verify (fri.py:201-319) on the transcripts the reference's Fri.prove wrote, read from their bytes, without a GPU: the verdict
    is the reference's (the disturbed codeword is rejected), and a stream with a wrong first root is rejected"""
    import stark_brainfuck_amd as sb
    from conftest import golden_bytes, load_golden
    rec = load_golden("fri.json")[tag]
    XF = sb.ExtensionField.main()
    BF = XF.modulus.coefficients[0].field
    fri = sb.Fri(BF.generator(), BF.primitive_nth_root(rec["N"]), rec["N"], rec["expansion"], rec["num_colinearity_tests"], XF)
    data = golden_bytes("fri_%s_stream.bin" % tag)
    vs = sb.ProofStream().deserialize(data)
    vs.read_index = rec["num_prepushed"]
    root0 = bytes.fromhex(rec["roots"][0])
    assert fri.verify(vs, root0) == rec["verify"]
    if rec["rounds"] > 1:
        bad = sb.ProofStream()
        bad.objects = list(vs.objects)
        bad.read_index = rec["num_prepushed"]
        bad.objects[rec["num_prepushed"]] = bytes(64)
        assert not fri.verify(bad, root0)
