"""BrainfuckStark.prove in a loop on a nested-loop program (argv: outer loop count, repetitions); prints the stage timing of each proof.
outer = 64 gives 37 254 cycles and an FRI domain of 2^22.
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stark_brainfuck_amd.brainfuck_stark import BrainfuckStark
from stark_brainfuck_amd.vm import VirtualMachine
outer = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
code = "+" * outer + "[>" + "+" * outer + "[>++++<-]<-]+++."
program = VirtualMachine.compile(code)
rt, inp, out = VirtualMachine.run(program)
m = VirtualMachine.simulate(program, input_data=inp)
for rep in range(reps):
    stark = BrainfuckStark(rt, len(m[1]), program, inp, out)
    stark.stage_timing = True
    t = time.perf_counter(); proof = stark.prove(program, *m); dt = time.perf_counter() - t
    print("prove %.1f ms" % (dt * 1e3), {k: round(v * 1e3, 2) for k, v in stark.timing.items()}, flush=True)
