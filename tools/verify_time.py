"""median latency of BrainfuckStark.verify on a Hello-World proof, native route and Python route (development tool)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stark_brainfuck_amd.brainfuck_stark import BrainfuckStark
from stark_brainfuck_amd.vm import VirtualMachine
code = "++++++++[>++++[>++>+++>+++>+<<<<-]>+>+>->>+[<]<-]>>.>---.+++++++..+++.>>.<-.<.+++.------.--------.>>+.>++."
program = VirtualMachine.compile(code)
rt, inp, out = VirtualMachine.run(program)
m = VirtualMachine.simulate(program, input_data=inp)
proof = BrainfuckStark(rt, len(m[1]), program, inp, out).prove(program, *m)
for mode in ("1", "0"):
    os.environ["BFS_NATIVE_VERIFY"] = mode
    ts = []
    for _ in range(60):
        t = time.perf_counter()
        assert BrainfuckStark(rt, len(m[1]), program, inp, out).verify(proof) is True
        ts.append(time.perf_counter() - t)
    ts = sorted(ts[5:])
    print("BFS_NATIVE_VERIFY=%s: median %.2f ms, best %.2f, worst %.2f (constructing the verifier object included)" % (mode, ts[len(ts) // 2] * 1e3, ts[0] * 1e3, ts[-1] * 1e3))
stark = BrainfuckStark(rt, len(m[1]), program, inp, out)
ts = []
for _ in range(60):
    t = time.perf_counter(); stark.verify(proof); ts.append(time.perf_counter() - t)
os.environ["BFS_NATIVE_VERIFY"] = "1"
ts = []
for _ in range(60):
    t = time.perf_counter(); stark.verify(proof); ts.append(time.perf_counter() - t)
ts = sorted(ts[5:]); print("native, one verifier object reused: median %.2f ms, best %.2f" % (ts[len(ts) // 2] * 1e3, ts[0] * 1e3))
