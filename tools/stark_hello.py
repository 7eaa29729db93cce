"""BrainfuckStark.prove on the "Hello World!" program (BASELINE.json config 4: FRI domain 2^17, 26 columns), timed."""
import os, sys, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stark_brainfuck_amd.brainfuck_stark import BrainfuckStark
from stark_brainfuck_amd.vm import VirtualMachine
from stark_brainfuck_amd.device import synchronize
code = sys.argv[1] if len(sys.argv) > 1 else "++++++++[>++++[>++>+++>+++>+<<<<-]>+>+>->>+[<]<-]>>.>---.+++++++..+++.>>.<-.<.+++.------.--------.>>+.>++."
t0 = time.perf_counter()
program = VirtualMachine.compile(code)
running_time, inp, out = VirtualMachine.run(program)
pm, mm, im, inm, om = VirtualMachine.simulate(program, input_data=inp)
t1 = time.perf_counter()
stark = BrainfuckStark(running_time, len(mm), program, inp, out)
stark.stage_timing = os.environ.get("STAGE_TIMING", "0") == "1"      # 1: synchronise after every stage (Python path); 0: the production path
t2 = time.perf_counter()
print("running time %d, memory rows %d, FRI domain 2^%d, setup %.3f s (vm %.3f s)" % (running_time, len(mm), stark.fri.domain.length.bit_length() - 1, t2 - t1, t1 - t0), flush=True)
reps = int(os.environ.get("REPS", "3"))
times = []
for rep in range(reps):
    t = time.perf_counter()
    proof = stark.prove(program, pm, mm, im, inm, om)
    synchronize()
    times.append(time.perf_counter() - t)
    if rep < 3:
        print("prove: %.2f ms, proof %d bytes, sha256 %s" % (times[-1] * 1e3, len(proof), hashlib.sha256(proof).hexdigest()[:16]), flush=True)
        if hasattr(stark, "timing"):
            print("   ", {k: round(v * 1e3, 3) for k, v in stark.timing.items()})
times = sorted(times[1:]) or times
print("median of %d: %.3f ms, best %.3f ms" % (len(times), times[len(times) // 2] * 1e3, times[0] * 1e3))
