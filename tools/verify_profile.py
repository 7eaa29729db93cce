"""cProfile of BrainfuckStark.verify on a Hello-World proof (development tool): where the verifier's 17 ms go."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stark_brainfuck_amd.brainfuck_stark import BrainfuckStark
from stark_brainfuck_amd.vm import VirtualMachine
code = "++++++++[>++++[>++>+++>+++>+<<<<-]>+>+>->>+[<]<-]>>.>---.+++++++..+++.>>.<-.<.+++.------.--------.>>+.>++."
program = VirtualMachine.compile(code)
rt, inp, out = VirtualMachine.run(program)
m = VirtualMachine.simulate(program, input_data=inp)
proof = BrainfuckStark(rt, len(m[1]), program, inp, out).prove(program, *m)
for _ in range(2):
    t = time.perf_counter()
    ok = BrainfuckStark(rt, len(m[1]), program, inp, out).verify(proof)
    print("verify", ok, "%.2f ms" % ((time.perf_counter() - t) * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    BrainfuckStark(rt, len(m[1]), program, inp, out).verify(proof)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(25)
st.sort_stats("cumulative").print_stats(35)
