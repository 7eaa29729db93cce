"""BrainfuckStark.prove + verify on nested-loop programs of growing running time (scale check beyond the goldens)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stark_brainfuck_amd.brainfuck_stark import BrainfuckStark
from stark_brainfuck_amd.vm import VirtualMachine
for outer in (8, 16, 32, 64, 128):
    # nested loops; the printed cell must stay below 256: the claim is about output CHARACTERS (vm.py:149), and the verifier
    # recomputes the output terminal from ord(character) (evaluation_argument.py:7-14)
    code = "+" * outer + "[>" + "+" * outer + "[>++++<-]<-]+++."
    program = VirtualMachine.compile(code)
    rt, inp, out = VirtualMachine.run(program)
    t = time.perf_counter()
    m = VirtualMachine.simulate(program, input_data=inp)
    t_vm = time.perf_counter() - t
    best, proof = None, None
    for rep in range(2):
        stark = BrainfuckStark(rt, len(m[1]), program, inp, out)
        stark.stage_timing = True
        t = time.perf_counter()
        proof = stark.prove(program, *m)
        dt = time.perf_counter() - t
        best = dt if best is None else min(best, dt)
    t = time.perf_counter()
    ok = BrainfuckStark(rt, len(m[1]), program, inp, out).verify(proof)
    tv = time.perf_counter() - t
    print("running time %6d  memory rows %6d  FRI domain 2^%d  trace %.3f s  prove %.3f s  verify %.3f s (%s)  proof %d bytes  %s" % (
        rt, len(m[1]), stark.fri.domain.length.bit_length() - 1, t_vm, best, tv, ok, len(proof),
        {k: round(v * 1e3, 1) for k, v in stark.timing.items() if v > 2e-3}), flush=True)
