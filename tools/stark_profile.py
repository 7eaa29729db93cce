"""cProfile of BrainfuckStark.prove on the Hello-World program (development tool): where the host time of a 6 ms proof goes."""
import cProfile
import os
import pstats
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stark_brainfuck_amd.brainfuck_stark import BrainfuckStark
from stark_brainfuck_amd.vm import VirtualMachine
code = "++++++++[>++++[>++>+++>+++>+<<<<-]>+>+>->>+[<]<-]>>.>---.+++++++..+++.>>.<-.<.+++.------.--------.>>+.>++."
program = VirtualMachine.compile(code)
running_time, inputs, outputs = VirtualMachine.run(program)
m = VirtualMachine.simulate(program, input_data=inputs)
for _ in range(3):
    BrainfuckStark(running_time, len(m[1]), program, inputs, outputs).prove(program, *m)
pr = cProfile.Profile()
starks = [BrainfuckStark(running_time, len(m[1]), program, inputs, outputs) for _ in range(20)]
pr.enable()
for s in starks:
    s.prove(program, *m)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(45)
