"""BrainfuckStark.prove on the nested-loop program (37 254 cycles, FRI domain 2^22) through the production path (native stage driver):
wall time per proof and the host's time per stage (development tool).  usage: python tools/stark_big_native.py [outer] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stark_brainfuck_amd.brainfuck_stark import BrainfuckStark
from stark_brainfuck_amd.vm import VirtualMachine
from stark_brainfuck_amd.device import synchronize
outer = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
code = "+" * outer + "[>" + "+" * outer + "[>++++<-]<-]+++."
program = VirtualMachine.compile(code)
rt, inp, out = VirtualMachine.run(program)
m = VirtualMachine.simulate(program, input_data=inp)
stark = BrainfuckStark(rt, len(m[1]), program, inp, out)
print("running time %d, rows %s, FRI domain 2^%d" % (rt, [len(x) for x in m], stark.fri.domain.length.bit_length() - 1))
for rep in range(reps):
    t = time.perf_counter(); proof = stark.prove(program, *m); synchronize(); dt = time.perf_counter() - t
    print("prove %.2f ms" % (dt * 1e3), {k: round(v * 1e3, 2) for k, v in stark.timing.items()}, flush=True)
