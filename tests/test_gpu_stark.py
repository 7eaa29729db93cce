"""BrainfuckStark.prove on the GPU against goldens captured from the reference's own prover:
tests/golden/stark_<name>.json + stark_<name>_proof.bin, made by tests/golden/gen_stark_golden.py with os.urandom
replaced by a SHAKE-256 stream.  Every stage is compared -- commitments, challenges, terminals, each quotient codeword,
degree bounds, the combination codeword, opened indices -- and finally the proof bytes themselves."""
import glob
import hashlib
import json
import os
import struct

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
NAMES = sorted(os.path.basename(p)[len("stark_"):-len(".json")] for p in glob.glob(os.path.join(GOLDEN, "stark_*.json")))


class Stream:
    """the byte stream the golden generator fed to the reference as os.urandom"""

    def __init__(self, tag):
        self.tag, self.pos, self.buf = tag, 0, b""

    def __call__(self, n):
        end = self.pos + n
        if end > len(self.buf):
            self.buf = hashlib.shake_256(b"bfs-golden-urandom" + self.tag).digest(max(2 * end, 1 << 16))
        out = self.buf[self.pos:end]
        self.pos = end
        return out


def sha_planes(a):
    """sha256 over elements in order, limbs interleaved (what gen_stark_golden.sha_elems hashes): a is (3, n) or (n,)"""
    a = np.ascontiguousarray(a.T if a.ndim == 2 else a, dtype="<u8")
    return hashlib.sha256(a.tobytes()).hexdigest()


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_prove_matches_reference(name, monkeypatch):
    from stark_brainfuck_amd import brainfuck_stark, salted_merkle, table
    from stark_brainfuck_amd.brainfuck_stark import BrainfuckStark
    from stark_brainfuck_amd.vm import VirtualMachine
    g = json.load(open(os.path.join(GOLDEN, "stark_%s.json" % name)))
    stream = Stream(name.encode())
    for mod in (brainfuck_stark, salted_merkle, table):
        monkeypatch.setattr(mod, "urandom", stream)

    program = VirtualMachine.compile(g["program"])
    assert [w.value for w in program] == g["compiled_program"]
    running_time, input_symbols, output_symbols = VirtualMachine.run(program, input_data=list(g["input"]))
    assert running_time == g["running_time"] and "".join(output_symbols) == g["output"]
    pm, mm, im, inm, om = VirtualMachine.simulate(program, input_data=list(input_symbols))
    stark = BrainfuckStark(running_time, len(mm), program, input_symbols, output_symbols)
    assert stark.max_degree == g["max_degree"] and stark.fri.domain.length == g["fri_domain_length"]
    assert [t.height for t in stark.tables] == g["table_heights"]

    stark.keep_intermediates = True
    proof = stark.prove(program, pm, mm, im, inm, om)
    last = stark._last
    n = stark.fri.domain.length
    assert stream.pos == g["urandom_bytes"], "the prover consumed a different amount of randomness"
    assert last["base_tree"].root().hex() == g["base_tree"]["root"]
    assert [list(c) for c in last["challenges"]] == g["quotients"][0]["challenges"]
    assert [list(t) for t in last["terminals"]] == g["terminals"]
    assert last["extension_tree"].root().hex() == g["extension_tree"]["root"]
    bounds, wrong = [], []
    for (buf, count), q in zip(last["quotient_buffers"], g["quotients"] + g["perm_quotients"]):
        host = buf.to_numpy(count * 3 * n).reshape(count, 3, n)
        want = q["sha"] if isinstance(q["sha"], list) else [q["sha"]]
        wrong += ["%s[%d]" % (q.get("table", "permutation argument"), k) for k in range(count) if sha_planes(host[k]) != want[k]]
        bounds += q["degree_bounds"] if "degree_bounds" in q else [q["degree_bound"]]
    assert not wrong, "quotient codewords that differ from the reference's: %s" % wrong
    assert last["quotient_degree_bounds"] == bounds
    assert last["weights_seed"].hex() == g["fiat_shamir"][1]["seed"], "transcript differs after the terminals"
    assert sha_planes(last["combination"].to_numpy()) == g["combination_tree"]["sha"]
    assert last["combination_tree"].root().hex() == g["combination_tree"]["root"]
    assert last["indices"] == g["indices"]
    assert len(proof) == g["proof_len"] and hashlib.sha256(proof).hexdigest() == g["proof_sha256"]
    path = os.path.join(GOLDEN, "stark_%s_proof.bin" % name)
    if os.path.exists(path):
        assert proof == open(path, "rb").read()


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_production_path_writes_the_reference_proof(name, monkeypatch):
    """the default prover (quotients folded into the combination in registers, bfs_air_combine; buffers handed back to the pool as it
    goes) on the reference's randomness: the proof must be the reference's, byte for byte -- and again on the same object"""
    from stark_brainfuck_amd import brainfuck_stark, salted_merkle, table
    from stark_brainfuck_amd.brainfuck_stark import BrainfuckStark
    from stark_brainfuck_amd.vm import VirtualMachine
    g = json.load(open(os.path.join(GOLDEN, "stark_%s.json" % name)))
    program = VirtualMachine.compile(g["program"])
    running_time, input_symbols, output_symbols = VirtualMachine.run(program, input_data=list(g["input"]))
    matrices = VirtualMachine.simulate(program, input_data=list(input_symbols))
    stark = BrainfuckStark(running_time, len(matrices[1]), program, input_symbols, output_symbols)
    assert stark.keep_intermediates is False
    for attempt in range(2):
        stream = Stream(name.encode())
        for mod in (brainfuck_stark, salted_merkle, table):
            monkeypatch.setattr(mod, "urandom", stream)
        proof = stark.prove(program, *matrices)
        assert stream.pos == g["urandom_bytes"]
        assert len(proof) == g["proof_len"] and hashlib.sha256(proof).hexdigest() == g["proof_sha256"], attempt
        assert "quotient_buffers" not in stark._last


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["loop", "two_io"])
def test_combination_by_row_windows_writes_the_reference_proof(name, monkeypatch):
    """bfs_zerofier_inverses_rows / bfs_air_combine_rows / bfs_difference_combine_rows (a cooperative proof's share of the pointwise
    stages) with the domain cut into ragged windows -- one point, an odd count, a piece that ends one short of the middle, the rest:
    the accumulator must come out as from one launch over the whole domain, i.e. the proof is still the reference's"""
    from stark_brainfuck_amd import brainfuck_stark, salted_merkle, table
    from stark_brainfuck_amd.brainfuck_stark import BrainfuckStark
    from stark_brainfuck_amd.vm import VirtualMachine
    g = json.load(open(os.path.join(GOLDEN, "stark_%s.json" % name)))
    program = VirtualMachine.compile(g["program"])
    running_time, input_symbols, output_symbols = VirtualMachine.run(program, input_data=list(g["input"]))
    matrices = VirtualMachine.simulate(program, input_data=list(input_symbols))
    stark = BrainfuckStark(running_time, len(matrices[1]), program, input_symbols, output_symbols)
    n = stark.fri.domain.length
    cuts = [0, 1, 258, n // 2 - 1, n]
    stark._row_windows = [(a, b - a) for a, b in zip(cuts, cuts[1:])]
    stream = Stream(name.encode())
    for mod in (brainfuck_stark, salted_merkle, table):
        monkeypatch.setattr(mod, "urandom", stream)
    proof = stark.prove(program, *matrices)
    assert len(proof) == g["proof_len"] and hashlib.sha256(proof).hexdigest() == g["proof_sha256"]


@pytest.mark.gpu
def test_combination_kernel_without_the_grouped_shift_pattern(monkeypatch):
    """air_combine_kernel<TABLE, GROUPED = false>: the launcher takes it when the degree shifts of a run of terms are NOT all equal,
    which sampled challenges practically never produce (round-3 advice).  Here the shifts are perturbed term by term through the
    prover's test hook, and the fused kernel's combination must equal what bfs_air_quotients + bfs_combination (one generic
    weighted sum over the written-out codewords, the keep_intermediates path) compute from the same shifts: same proof bytes."""
    import numpy as np
    from stark_brainfuck_amd import brainfuck_stark, salted_merkle, table
    from stark_brainfuck_amd.brainfuck_stark import BrainfuckStark
    from stark_brainfuck_amd.vm import VirtualMachine
    g = json.load(open(os.path.join(GOLDEN, "stark_two_io.json")))
    program = VirtualMachine.compile(g["program"])
    running_time, input_symbols, output_symbols = VirtualMachine.run(program, input_data=list(g["input"]))
    matrices = VirtualMachine.simulate(program, input_data=list(input_symbols))

    def tweak(shifts):
        return shifts + (np.arange(len(shifts), dtype=np.uint64) * np.uint64(7)) % np.uint64(5)      # neighbours differ: no run is uniform
    proofs = {}
    for keep in (True, False):
        stark = BrainfuckStark(running_time, len(matrices[1]), program, input_symbols, output_symbols)
        stark.keep_intermediates = keep
        stark._shift_tweak = tweak
        stream = Stream(b"two_io")
        for mod in (brainfuck_stark, salted_merkle, table):
            monkeypatch.setattr(mod, "urandom", stream)
        proofs[keep] = stark.prove(program, *matrices)
    assert proofs[True] == proofs[False]
    assert hashlib.sha256(proofs[False]).hexdigest() != g["proof_sha256"]          # (the tweak did change the combination)


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_verify_accepts_reference_proofs_and_rejects_tampering(name):
    """the verifier mirror (brainfuck_stark.py:343-579) on proofs written by the reference itself"""
    from stark_brainfuck_amd.brainfuck_stark import BrainfuckStark
    from stark_brainfuck_amd.vm import VirtualMachine
    g = json.load(open(os.path.join(GOLDEN, "stark_%s.json" % name)))
    program = VirtualMachine.compile(g["program"])
    running_time, input_symbols, output_symbols = VirtualMachine.run(program, input_data=list(g["input"]))
    _, mm, _, _, _ = VirtualMachine.simulate(program, input_data=list(input_symbols))
    proof = open(os.path.join(GOLDEN, "stark_%s_proof.bin" % name), "rb").read()
    stark = BrainfuckStark(running_time, len(mm), program, input_symbols, output_symbols)
    assert stark.verify(proof) is True
    # a claim about a different output must fail (the output evaluation terminal no longer matches, evaluation_argument.py)
    other = BrainfuckStark(running_time, len(mm), program, input_symbols, list(output_symbols) + ["!"])
    try:
        assert other.verify(proof) is False
    except AssertionError:
        pass
    # flipping one bit inside an opened digest breaks an authentication path
    pos = proof.index(bytes.fromhex(g["combination_tree"]["root"])) + 200
    bad = bytearray(proof)
    bad[pos] ^= 1
    try:
        assert stark.verify(bytes(bad)) is False
    except (AssertionError, Exception):
        pass


@pytest.mark.gpu
def test_prove_then_verify_fresh_randomness():
    """with real os.urandom there is no golden: the proof must verify, and two proofs of the same claim must differ"""
    from stark_brainfuck_amd.brainfuck_stark import BrainfuckStark
    from stark_brainfuck_amd.vm import VirtualMachine
    program = VirtualMachine.compile(",>,<[->+<]>.")
    running_time, input_symbols, output_symbols = VirtualMachine.run(program, input_data=list("!#"))
    assert "".join(output_symbols) == "D"
    pm, mm, im, inm, om = VirtualMachine.simulate(program, input_data=list(input_symbols))
    stark = BrainfuckStark(running_time, len(mm), program, input_symbols, output_symbols)
    proof_a = stark.prove(program, pm, mm, im, inm, om)
    proof_b = BrainfuckStark(running_time, len(mm), program, input_symbols, output_symbols).prove(program, pm, mm, im, inm, om)
    assert proof_a != proof_b
    assert BrainfuckStark(running_time, len(mm), program, input_symbols, output_symbols).verify(proof_a) is True
    assert BrainfuckStark(running_time, len(mm), program, input_symbols, output_symbols).verify(proof_b) is True


def _debug_cases():
    path = os.path.join(GOLDEN, "debug_checks.json")
    return json.load(open(path))["cases"] if os.path.exists(path) else []


@pytest.mark.gpu
@pytest.mark.parametrize("case", _debug_cases(), ids=lambda c: c["tag"])
def test_debug_degree_checks_stop_where_the_reference_stops(case, monkeypatch):
    """DEBUG mode (stark_brainfuck_amd/debug_checks.py; brainfuck_stark.py:251-290, table.py:170-176 / 219-234 / 264-284): with `DEBUG` in
    the environment prove() interpolates every quotient and every term of the combination and asserts the degrees the reference asserts.
    tests/golden/debug_checks.json (gen_debug_golden.py) holds what the REFERENCE did, under DEBUG=1, with a clean trace and with one cell
    of a trace matrix changed: a clean trace passes every check and gives the proof the reference gave; a corrupted one is stopped by an
    AssertionError in the same function, for the same table, at the same constraint."""
    from stark_brainfuck_amd import brainfuck_stark, salted_merkle, table
    from stark_brainfuck_amd.brainfuck_stark import BrainfuckStark
    from stark_brainfuck_amd.vm import VirtualMachine
    g = json.load(open(os.path.join(GOLDEN, "debug_checks.json")))
    P = (1 << 64) - (1 << 32) + 1
    program = VirtualMachine.compile(g["program"])
    running_time, input_symbols, output_symbols = VirtualMachine.run(program, input_data=[])
    matrices = dict(zip(("processor", "memory", "instruction", "input", "output"), VirtualMachine.simulate(program, input_data=[])))
    assert {k: [m.values.shape[0], m.values.shape[1] if m.values.shape[0] else 0] for k, m in matrices.items()} == g["shapes"]
    if case["matrix"]:
        v = matrices[case["matrix"]].values
        v[case["row"], case["column"]] = (int(v[case["row"], case["column"]]) + case["add"]) % P
    stream = Stream(("debug-" + case["tag"]).encode())
    for mod in (brainfuck_stark, salted_merkle, table):
        monkeypatch.setattr(mod, "urandom", stream)
    monkeypatch.setenv("DEBUG", "1")
    stark = BrainfuckStark(running_time, len(matrices["memory"]), program, input_symbols, output_symbols)
    assert stark.fri.domain.length == g["fri_domain_length"]
    args = (program, matrices["processor"], matrices["memory"], matrices["instruction"], matrices["input"], matrices["output"])
    if case["outcome"] == "passed":
        proof = stark.prove(*args)
        assert stream.pos == case["urandom_bytes"]
        assert len(proof) == case["proof_len"] and hashlib.sha256(proof).hexdigest() == case["proof_sha256"]
        assert stark.keep_intermediates is False and "quotient_buffers" not in stark._last
        monkeypatch.delenv("DEBUG")
        assert stark.verify(proof) is True
        return
    with pytest.raises(AssertionError) as info:
        stark.prove(*args)
    function, table_name, index = info.value.where
    assert function == case["function"], (info.value.where, case)
    if "table" in case:
        assert table_name == case["table"]
    # which constraint: the loop variable of the reference's frame -- `qc`'s position in boundary_quotients (its `l` is left over from the loop
    # that made the codewords), `l` in transition_quotients, `i` in terminal_quotients
    want = case.get("index_qc") if function == "boundary_quotients" else case.get("index_l", case.get("index_i"))
    if function != "prove" and want is not None:
        assert index == want, (info.value.where, case)
    # without DEBUG the same trace goes through (the prover does not check its witness), and the proof is rejected
    monkeypatch.delenv("DEBUG")
    stream2 = Stream(("debug-" + case["tag"]).encode())
    for mod in (brainfuck_stark, salted_merkle, table):
        monkeypatch.setattr(mod, "urandom", stream2)
    bad = BrainfuckStark(running_time, len(matrices["memory"]), program, input_symbols, output_symbols).prove(*args)
    try:
        assert BrainfuckStark(running_time, len(matrices["memory"]), program, input_symbols, output_symbols).verify(bad) is False
    except AssertionError:
        pass


WRAPPING_PROGRAMS = [
    "[->+<][>]+>[-]+++++[>+[>+<-]<-]-++++.----",      # the one tools/soak_stark.py found in round 4 (cells pass through p - 1)
    "-+.", "--++.", "->-<+.>++.", "+[-]-[+]+.", "-[>-<+]>+.", "++[>--<-]>++++.", "-->-<[>+<+]>+++.", "->->-<<+[>+>+<<-]>>+.",
]


@pytest.mark.gpu
@pytest.mark.parametrize("code", WRAPPING_PROGRAMS)
def test_traces_full_of_values_next_to_zero_and_p_prove_and_verify(code, monkeypatch):
    """trace columns are counters, zeros and p - 1: the values for which an unreduced butterfly sum really exceeds p.  Round 4: a
    first version of the NTT's lazy sums passed every random-data comparison and all ten reference proofs, and wrote a codeword
    value >= p into the proof of the first program here (found by tools/soak_stark.py; verify() rejected it).  Both prover paths
    must write the same bytes, verify() -- host code that shares no kernel with the prover -- must accept them, and every integer
    in the proof must be a canonical residue.  (In that proof the value >= p sat at an unopened position; what broke was downstream
    arithmetic that relies on canonical operands.  The NTT's own outputs on such inputs are compared with the oracle bit for bit in
    tests/test_gpu_parity.py::test_ntt_on_values_next_to_zero_and_p.)"""
    from stark_brainfuck_amd import brainfuck_stark, salted_merkle, table
    from stark_brainfuck_amd.brainfuck_stark import BrainfuckStark
    from stark_brainfuck_amd.vm import VirtualMachine
    import pickle
    program = VirtualMachine.compile(code)
    matrices = VirtualMachine.simulate(program, input_data=[], max_cycles=5000)
    rt = len(matrices[0])
    outputs = [chr(int(v) % 256) for v in matrices[4].values.reshape(-1)]
    proofs = []
    for keep in (False, True):
        stream = Stream(code.encode())
        for mod in (brainfuck_stark, salted_merkle, table):
            monkeypatch.setattr(mod, "urandom", stream)
        stark = BrainfuckStark(rt, len(matrices[1]), program, [], outputs)
        stark.keep_intermediates = keep
        proofs.append(stark.prove(program, *matrices))
    assert proofs[0] == proofs[1]
    assert BrainfuckStark(rt, len(matrices[1]), program, [], outputs).verify(proofs[0]) is True
    P = (1 << 64) - (1 << 32) + 1

    def integers(o):
        if isinstance(o, bool) or type(o).__name__ in ("BaseField", "ExtensionField"):          # (a field object holds p itself)
            return
        if isinstance(o, int):
            yield o
        elif isinstance(o, (list, tuple)):
            for c in o:
                yield from integers(c)
        elif hasattr(o, "__dict__"):
            for c in vars(o).values():
                yield from integers(c)
        elif hasattr(o, "__slots__"):
            for name in o.__slots__:
                yield from integers(getattr(o, name, None))
    from stark_brainfuck_amd.ip import ProofStream
    objects = ProofStream().deserialize(proofs[0]).objects
    values = [v for v in integers(objects)]
    assert len(values) > 100 and all(0 <= v < P for v in values)


@pytest.mark.gpu
def test_stage_timing_changes_the_timing_not_the_proof(monkeypatch):
    """BrainfuckStark.stage_timing = True synchronises the stream after every stage (bench.py's breakdown_ms); the default lets the
    stages overlap.  Same proof bytes either way, and `timing` names the same stages"""
    from stark_brainfuck_amd import brainfuck_stark, salted_merkle, table
    from stark_brainfuck_amd.brainfuck_stark import BrainfuckStark
    from stark_brainfuck_amd.vm import VirtualMachine
    program = VirtualMachine.compile("++>+++[<+>-]<.")
    rt, inputs, outputs = VirtualMachine.run(program)
    matrices = VirtualMachine.simulate(program, input_data=inputs)
    proofs, stages = [], []
    for staged in (False, True):
        stream = Stream(b"stage-timing")
        for mod in (brainfuck_stark, salted_merkle, table):
            monkeypatch.setattr(mod, "urandom", stream)
        stark = BrainfuckStark(rt, len(matrices[1]), program, inputs, outputs)
        stark.stage_timing = staged
        proofs.append(stark.prove(program, *matrices))
        stages.append(list(stark.timing))
        assert all(v >= 0 for v in stark.timing.values())
    # (the unstaged proof takes the native stage driver, which also reports the host time in front of and between its two calls)
    assert proofs[0] == proofs[1] and [s for s in stages[0] if not s.startswith("host_")] == stages[1]
    assert {"base_lde", "base_tree", "ext_tree", "combination", "fri"} <= set(stages[0])
    assert BrainfuckStark.stage_timing is False


@pytest.mark.gpu
def test_prove_and_verify_with_other_protocol_parameters():
    """expansion factor 16 with 32 colinearity checks and 128 opened indices (the reference hard-codes 4 / 1 / 2 "for speed",
    brainfuck_stark.py:31-36; FRI caps the number of checks at the length of the last codeword, fri.py:69-70)"""
    from stark_brainfuck_amd.brainfuck_stark import BrainfuckStark
    from stark_brainfuck_amd.vm import VirtualMachine
    program = VirtualMachine.compile("++>+++<[->+<]>.")
    running_time, inputs, outputs = VirtualMachine.run(program)
    matrices = VirtualMachine.simulate(program, input_data=inputs)
    args = (running_time, len(matrices[1]), program, inputs, outputs)
    stark = BrainfuckStark(*args, log_expansion_factor=4, security_level=128)
    assert stark.expansion_factor == 16 and stark.num_colinearity_checks == 32
    proof = stark.prove(program, *matrices)
    assert BrainfuckStark(*args, log_expansion_factor=4, security_level=128).verify(proof) is True
    try:
        accepted = BrainfuckStark(*args).verify(proof)          # the default parameters describe a different protocol
    except Exception:
        accepted = False
    assert accepted is False


@pytest.mark.gpu
def test_prover_objects_release_their_memory_without_the_garbage_collector():
    """no reference cycles through the HBM buffers: dropping a prover that kept its intermediates (trees, quotient codewords)
    gives every block back to the pool at once -- prove() runs under gc.freeze(), where cyclic garbage would wait indefinitely"""
    import gc
    from stark_brainfuck_amd import device
    from stark_brainfuck_amd.brainfuck_stark import BrainfuckStark
    from stark_brainfuck_amd.vm import VirtualMachine
    program = VirtualMachine.compile("++[>+<-]>.")
    running_time, inputs, outputs = VirtualMachine.run(program)
    matrices = VirtualMachine.simulate(program, input_data=inputs)
    gc.collect()
    device.synchronize()
    live_before = device.pool_stats()[0]
    gc.disable()
    try:
        for keep in (True, False):
            stark = BrainfuckStark(running_time, len(matrices[1]), program, inputs, outputs)
            stark.keep_intermediates = keep
            proof = stark.prove(program, *matrices)
            if keep:
                assert device.pool_stats()[0] > live_before          # trees and quotient codewords are alive
            del stark
            assert device.pool_stats()[0] == live_before, "keep_intermediates = %s" % keep
    finally:
        gc.enable()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["two_io", "loop"])
def test_prove_from_plain_lists_of_elements(name, monkeypatch):
    """drop-in use: the matrices are plain Python lists of rows of BaseFieldElement, as the reference's VirtualMachine.simulate returns
    them (no arrays attached, every element an object) -- the proof is still the reference's"""
    from stark_brainfuck_amd import brainfuck_stark, salted_merkle, table
    from stark_brainfuck_amd.brainfuck_stark import BrainfuckStark
    from stark_brainfuck_amd.vm import VirtualMachine
    g = json.load(open(os.path.join(GOLDEN, "stark_%s.json" % name)))
    program = VirtualMachine.compile(g["program"])
    running_time, input_symbols, output_symbols = VirtualMachine.run(program, input_data=list(g["input"]))
    matrices = [[list(row) for row in m] for m in VirtualMachine.simulate_objects(program, input_data=list(input_symbols))]
    assert all(type(m) is list for m in matrices)
    stream = Stream(name.encode())
    for mod in (brainfuck_stark, salted_merkle, table):
        monkeypatch.setattr(mod, "urandom", stream)
    stark = BrainfuckStark(running_time, len(matrices[1]), program, input_symbols, output_symbols)
    proof = stark.prove(program, *matrices)
    assert hashlib.sha256(proof).hexdigest() == g["proof_sha256"]


@pytest.mark.gpu
@pytest.mark.parametrize("code, inputs", [("++>+++[<+>-]<.", ""), (",.,.", "ab"), ("+++[>+++[>++<-]<-]>>.", "")])
def test_one_transform_for_all_tables_equals_one_per_table(code, inputs, monkeypatch):
    """table.lde_tables (every table interpolated into its rows of one coefficient buffer, ONE coset transform over the columns of all
    tables) against Table.lde / Table.ldex called table by table (table.py:112-148 of the reference does the latter): same
    randomizers in the same order, so the codewords, the trace columns kept in HBM and the support summaries must be identical --
    also for tables of height zero (no input / no output)."""
    from stark_brainfuck_amd import table
    from stark_brainfuck_amd.brainfuck_stark import BrainfuckStark
    from stark_brainfuck_amd.table import extend_tables_device, lde_tables
    from stark_brainfuck_amd.vm import VirtualMachine
    program = VirtualMachine.compile(code)
    running_time, input_symbols, output_symbols = VirtualMachine.run(program, input_data=list(inputs))
    challenges = tuple((3 + 7 * k + (1 << 40), 5 + k + (1 << 33), 11 * k + (1 << 35)) for k in range(11))
    initials = [(1 << 34, 3, 1 << 50), (1 << 41, 9, 1 << 36)]
    results = []
    for batched in (False, True):
        monkeypatch.setattr(table, "urandom", Stream(b"lde-" + code.encode()))
        matrices = VirtualMachine.simulate(program, input_data=list(input_symbols))
        stark = BrainfuckStark(running_time, len(matrices[1]), program, input_symbols, output_symbols)
        for t, m in zip(stark.tables, (matrices[0], matrices[2], matrices[1], matrices[3], matrices[4])):
            t.matrix = m
        for t in (stark.processor_table, stark.memory_table, stark.instruction_table, stark.input_table, stark.output_table):
            t.pad()
        domain, n = stark.fri.domain, stark.fri.domain.length
        if batched:
            lde_tables(stark.tables, domain)
        else:
            for t in stark.tables:
                t.lde(domain)
        base = [t.base_codewords.to_numpy(t.base_width * n) for t in stark.tables]
        kept = [t._base_device.to_numpy() if t._base_device is not None else None for t in stark.tables]
        extend_tables_device(stark.tables, challenges, initials)
        if batched:
            lde_tables(stark.tables, domain, extension=True)
        else:
            for t in stark.tables:
                t.ldex(domain)
        ext = [t.ext_codewords.to_numpy(3 * (t.full_width - t.base_width) * n) for t in stark.tables]
        results.append((base, kept, ext, [t.ext_sharing_moduli(n) for t in stark.tables], [t.height for t in stark.tables]))
    (base_a, kept_a, ext_a, moduli_a, heights), (base_b, kept_b, ext_b, moduli_b, _) = results
    assert 0 in heights or inputs        # at least the input table is empty in the programs without input
    for k in range(5):
        assert np.array_equal(base_a[k], base_b[k]), k
        assert np.array_equal(ext_a[k], ext_b[k]), k
        assert (kept_a[k] is None) == (kept_b[k] is None) and (kept_a[k] is None or np.array_equal(kept_a[k], kept_b[k]))
    assert moduli_a == moduli_b
    assert any(a.any() for a in base_a) and any(a.any() for a in ext_a)


@pytest.mark.gpu
def test_trace_padding_on_the_device():
    """bfs_trace_pad (csrc/scan.hip): the trace tables go up as the virtual machine wrote them (row-major, unpadded) and are padded,
    transposed and reduced on the device, with the scan masks of their extensions.  Against the reference's rules restated here in
    numpy -- Table.pad: processor_table.py:24-35 (the cycle count keeps counting; ip, mp, mv, mvi stay), instruction_table.py:19-25
    (the last address repeats), memory_table.py:40-44 (dummy rows: cycle counting up, mp / mv kept, dummy = 1), io_table.py:17-21
    (zero rows) -- for real traces and for synthetic rows with residues >= p, tables without rows, with one row, and with exactly a
    power of two of rows."""
    import ctypes
    from stark_brainfuck_amd import _lib
    from stark_brainfuck_amd.device import DeviceBuffer
    from stark_brainfuck_amd.vm import VirtualMachine
    lib = _lib.load()
    P = (1 << 64) - (1 << 32) + 1
    WIDTH = [7, 3, 4, 1, 1]

    def npo2(k):
        h = 1
        while h < k:
            h <<= 1
        return h if k else 0

    def expect(kind, rows, width, height):
        k = rows.shape[0]
        vals = [[int(v) % P for v in r[:width]] for r in rows]
        out = [[0] * height for _ in range(width)]
        for r in range(k):
            for c in range(width):
                out[c][r] = vals[r][c]
        last = vals[-1] if k else [0] * width
        for r in range(k, height):
            j = r - k + 1
            if kind == 0:
                out[0][r] = (last[0] + j) % P
                for c in (1, 4, 5, 6):
                    out[c][r] = last[c]
            elif kind == 1:
                out[0][r] = last[0]
            elif kind == 2:
                out[0][r], out[1][r], out[2][r], out[3][r] = (last[0] + j) % P, last[1], last[2], 1
        masks = []
        if kind == 0:
            masks = [[int(v != 0) for v in out[2]], [int(v == ord(",")) for v in out[2]], [int(v == ord(".")) for v in out[2]]]
        elif kind == 1:
            same = [r > 0 and out[0][r] == out[0][r - 1] for r in range(height)]
            masks = [[int(out[1][r] != 0 and same[r]) for r in range(height)], [int(not s) for s in same]]
        elif kind == 2:
            masks = [[int(v == 0) for v in out[3]]]
        return out, masks

    def check(tables, heights):
        """tables: five (rows x stride) uint64 arrays"""
        bufs, outs, masks, structs = [], [], [], (_lib.TracePadTable * 5)()
        for t, (rows, h) in enumerate(zip(tables, heights)):
            raw = DeviceBuffer.from_numpy(np.ascontiguousarray(rows).reshape(-1)) if rows.size else None
            out = DeviceBuffer(max(WIDTH[t] * h, 1))
            mk = [DeviceBuffer(max((h + 7) // 8, 1)) for _ in range(3)]
            bufs.append(raw)
            outs.append(out)
            masks.append(mk)
            s = structs[t]
            s.d_rows, s.rows, s.row_stride, s.height = (raw.ptr if raw else None), rows.shape[0], (rows.shape[1] if rows.size else 0), h
            s.d_out, s.d_mask0, s.d_mask1, s.d_mask2, s.kind, s.width = out.ptr, mk[0].ptr, mk[1].ptr, mk[2].ptr, t, WIDTH[t]
        _lib.check(lib.bfs_trace_pad(structs, 5, 0))
        for t, (rows, h) in enumerate(zip(tables, heights)):
            want, want_masks = expect(t, rows, WIDTH[t], h)
            if h:
                got = outs[t].to_numpy(WIDTH[t] * h).reshape(WIDTH[t], h)
                assert got.tolist() == want, "table %d" % t
            for k, wm in enumerate(want_masks):
                got = masks[t][k].to_numpy((h + 7) // 8).view(np.uint8)[:h] if h else np.zeros(0, dtype=np.uint8)
                assert got.tolist() == wm, "table %d mask %d" % (t, k)

    # real traces: a program that reads and writes, one that does neither
    for code, inp in ((",[.,]", "hello\x00"), ("++[>+++<-]>.", ""), ("+", "")):
        program = VirtualMachine.compile(code)
        pm, mm, im, inm, om = VirtualMachine.simulate(program, input_data=inp)
        tables = [np.ascontiguousarray(m.values) if len(m) else np.zeros((0, w), dtype=np.uint64) for m, w in zip((pm, im, mm, inm, om), WIDTH)]
        check(tables, [npo2(len(t)) for t in tables])
    # synthetic rows: residues next to 0 and p and beyond p, wider rows than the table uses, every shape of (rows, height)
    rng = np.random.default_rng(77)
    edge = np.array([0, 1, 44, 46, P - 1, P, P + 5, (1 << 64) - 1], dtype=np.uint64)
    for shape in ([0, 0, 0, 0, 0], [1, 1, 1, 1, 1], [256, 128, 512, 2, 4], [300, 5, 1000, 3, 0], [70000, 70001, 140000, 0, 1]):
        tables = []
        for t, k in enumerate(shape):
            stride = WIDTH[t] + int(rng.integers(0, 3))
            rows = rng.integers(0, 1 << 63, (k, stride), dtype=np.uint64)
            pick = rng.integers(0, 3, (k, stride))
            rows = np.where(pick == 0, edge[rng.integers(0, len(edge), (k, stride))], rows)
            if t == 1 and k:
                rows[:, 0] = np.sort(rng.integers(0, max(k // 3, 1), k).astype(np.uint64))      # repeated addresses, as in a real table
            tables.append(rows)
        check(tables, [npo2(k) for k in shape])
        check(tables, [2 * npo2(k) if k else 4 for k in shape])          # taller than the rows ask for (IO tables: zero rows)


@pytest.mark.gpu
def test_proving_in_a_loop_does_not_grow_the_process(monkeypatch):
    """a prover that runs for hours must not depend on Python's cyclic collector to give memory back: with the collector switched
    off, 150 proofs + verifications -- production path and Python stages, with the operating system's randomness and with a replaced
    urandom (explicit 24 n-byte salt buffers) -- leave the resident set where it was.  (Round 5: `ctypes.cast(buffer, c_void_p)` makes
    the buffer part of a reference cycle, and a nested function that calls itself held every transcript; a soak grew by 5 MB per proof
    and lost its box after twenty minutes.  tools/leak_probe.py is the long form of this test.)"""
    import gc
    from stark_brainfuck_amd import brainfuck_stark, salted_merkle, table
    from stark_brainfuck_amd.brainfuck_stark import BrainfuckStark
    from stark_brainfuck_amd.vm import VirtualMachine
    code = "++++[>++++[>++<-]<-]>>."
    program = VirtualMachine.compile(code)
    rt, inp, out = VirtualMachine.run(program)
    m = VirtualMachine.simulate(program, input_data=inp)

    class Stream:
        def __init__(self, tag):
            self.tag, self.pos, self.buf = tag, 0, b""

        def __call__(self, n):
            end = self.pos + n
            if end > len(self.buf):
                self.buf = hashlib.shake_256(b"loop" + self.tag).digest(max(2 * end, 1 << 16))
            o = self.buf[self.pos:end]
            self.pos = end
            return o

    def rss():
        return int(open("/proc/self/statm").read().split()[1]) * os.sysconf("SC_PAGE_SIZE") / 2**20

    def one(k):
        keep, replaced = bool(k & 1), bool(k & 2)
        if replaced:
            s = Stream(str(k).encode())
            for mod in (brainfuck_stark, salted_merkle, table):
                monkeypatch.setattr(mod, "urandom", s)
        else:
            monkeypatch.undo()
        stark = BrainfuckStark(rt, len(m[1]), program, inp, out)
        stark.keep_intermediates = keep
        proof = stark.prove(program, *m)
        assert BrainfuckStark(rt, len(m[1]), program, inp, out).verify(proof) is True

    for k in range(8):              # pools, caches, templates: everything that is made once
        one(k)
    gc.collect()
    gc.disable()
    try:
        before = rss()
        for k in range(150):
            one(k)
        grown = rss() - before
    finally:
        gc.enable()
        monkeypatch.undo()
    assert grown < 24, "150 proofs grew the process by %.0f MiB with the cyclic collector off" % grown
