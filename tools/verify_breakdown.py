"""where the native verifier's time goes (development tool): bytes -> native graph, begin, Python between, finish"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stark_brainfuck_amd import _lib
from stark_brainfuck_amd.brainfuck_stark import BrainfuckStark
from stark_brainfuck_amd.ip import NativeTranscript
from stark_brainfuck_amd.vm import VirtualMachine
code = "++++++++[>++++[>++>+++>+++>+<<<<-]>+>+>->>+[<]<-]>>.>---.+++++++..+++.>>.<-.<.+++.------.--------.>>+.>++."
program = VirtualMachine.compile(code)
rt, inp, out = VirtualMachine.run(program)
m = VirtualMachine.simulate(program, input_data=inp)
proof = BrainfuckStark(rt, len(m[1]), program, inp, out).prove(program, *m)
lib = _lib.load()
acc = {}
def lap(name, t0):
    acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
N = 100
for _ in range(N):
    t0 = time.perf_counter(); stark = BrainfuckStark(rt, len(m[1]), program, inp, out); lap("construct", t0)
    t0 = time.perf_counter(); t = NativeTranscript.from_bytes(proof); lap("bfs_ps_loads", t0)
    t0 = time.perf_counter(); del t; lap("free", t0)
    t0 = time.perf_counter(); assert stark.verify(proof); lap("verify total", t0)
print({k: round(v / N * 1e3, 3) for k, v in acc.items()}, "ms; proof bytes", len(proof))
