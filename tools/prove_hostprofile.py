"""Where the HOST time of BrainfuckStark.prove goes on the Hello-World program (config 4): per-stage host time without stage
synchronisation, then cProfile over many proofs (development tool).  usage: python tools/prove_hostprofile.py [proofs]"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stark_brainfuck_amd.brainfuck_stark import BrainfuckStark
from stark_brainfuck_amd.vm import VirtualMachine
from stark_brainfuck_amd.device import synchronize
code = "++++++++[>++++[>++>+++>+++>+<<<<-]>+>+>->>+[<]<-]>>.>---.+++++++..+++.>>.<-.<.+++.------.--------.>>+.>++."
program = VirtualMachine.compile(code)
rt, inp, out = VirtualMachine.run(program)
m = VirtualMachine.simulate(program, input_data=inp)
stark = BrainfuckStark(rt, len(m[1]), program, inp, out)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for _ in range(30):
    stark.prove(program, *m)
synchronize()
acc, t0 = {}, time.perf_counter()
for _ in range(reps):
    stark.prove(program, *m)
    for k, v in stark.timing.items():
        acc[k] = acc.get(k, 0.0) + v
synchronize()
wall = (time.perf_counter() - t0) / reps
print("prove: %.3f ms per proof (no stage synchronisation); host time per stage (us):" % (wall * 1e3))
print("   ", {k: round(v / reps * 1e6, 1) for k, v in acc.items()})
pr = cProfile.Profile()
pr.enable()
for _ in range(reps):
    stark.prove(program, *m)
synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(60)
print(s.getvalue()[:14000])
