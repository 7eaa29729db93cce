"""Randomised soak of the STARK prover: random terminating Brainfuck programs with random inputs.  For each one
  * the production path (quotients folded into the combination, device scans) and the keep_intermediates path (quotient codewords
    written out) must produce the SAME proof from the same randomness,
  * verify() -- an independent host implementation of the constraints (air.evaluate) -- must accept it, and reject it for a
    different claimed output.
usage: python tools/soak_stark.py [seconds] [seed]"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stark_brainfuck_amd import brainfuck_stark, salted_merkle, table
from stark_brainfuck_amd.brainfuck_stark import BrainfuckStark
from stark_brainfuck_amd.vm import VirtualMachine


class Stream:
    def __init__(self, tag):
        self.tag, self.pos, self.buf = tag, 0, b""

    def __call__(self, n):
        end = self.pos + n
        if end > len(self.buf):
            self.buf = hashlib.shake_256(b"soak" + self.tag).digest(max(2 * end, 1 << 16))
        out = self.buf[self.pos:end]
        self.pos = end
        return out


def random_program(rng):
    out = []
    for _ in range(int(rng.integers(1, 30))):
        k = rng.integers(0, 12)
        if k < 4:
            out.append("+-"[rng.integers(0, 2)] * int(rng.integers(1, 5)))
        elif k < 6:
            out.append("><"[rng.integers(0, 2)])
        elif k == 6:
            out.append(",")
        elif k == 7:
            out.append(".")
        elif k == 8:
            out.append("[-]")
        elif k == 9:
            out.append("[->+<]")
        elif k == 10:
            out.append("+" * int(rng.integers(1, 6)) + "[>" + "+" * int(rng.integers(1, 6)) + "[>+<-]<-]")
        else:
            out.append("[>]")
    return "".join(out)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    t0, count, shapes = time.time(), 0, set()
    last_log = t0
    while time.time() - t0 < budget:
        code = random_program(rng)
        inp = [chr(int(c)) for c in rng.integers(1, 127, code.count(",") + 3)]
        program = VirtualMachine.compile(code)
        try:        # the native machine with a cycle limit: `-[-]` counts down from p - 1, which run() would follow to the end
            matrices = VirtualMachine.simulate(program, input_data=inp, max_cycles=20000)
        except AssertionError:                  # limit reached / input exhausted
            continue
        if len(matrices[4]) and (matrices[4]._ids == 1).any():
            continue                            # `.` on a cell that was never written: KeyError in the reference's run() (vm.py:149)
        rt = len(matrices[0])
        inputs = inp[:len(matrices[3])]         # the claim is about the symbols the program actually read
        outputs = [chr(int(v) % 256) for v in matrices[4].values.reshape(-1)]
        if len(matrices[4]) and int(matrices[4].values.max()) >= 256:
            continue      # the claim is about output CHARACTERS (value % 256, vm.py:149) while the table holds the value: no valid proof
        if int(matrices[0].values[:, 4].max()) >> 32:
            continue      # the memory pointer wrapped below zero: the visited addresses are no longer contiguous, which the
                          # reference's memory AIR requires (memory_table.py:56-60) -- such a trace has no valid proof there either
        proofs = []
        for keep in (False, True):
            stream = Stream(code.encode())
            for mod in (brainfuck_stark, salted_merkle, table):
                mod.urandom = stream
            stark = BrainfuckStark(rt, len(matrices[1]), program, inputs, outputs)
            stark.keep_intermediates = keep
            proofs.append(stark.prove(program, *matrices))
        assert proofs[0] == proofs[1], "the two prover paths disagree on %r" % code
        assert BrainfuckStark(rt, len(matrices[1]), program, inputs, outputs).verify(proofs[0]) is True, "rejected: %r" % code
        try:
            wrong = BrainfuckStark(rt, len(matrices[1]), program, inputs, list(outputs) + ["!"]).verify(proofs[0])
        except AssertionError:
            wrong = False
        assert wrong is False, "accepted a false claim: %r" % code
        count += 1
        shapes.add(tuple(t.height for t in stark.tables) + (stark.fri.domain.length,))
        if os.environ.get("SOAK_MEMLOG") and time.time() - last_log >= 30:
            # resident set of this process and free device memory: a soak must not grow either without bound
            last_log = time.time()
            rss = int(open("/proc/self/statm").read().split()[1]) * os.sysconf("SC_PAGE_SIZE") / 2**20
            try:
                import torch
                free, total = torch.cuda.mem_get_info()
                dev = "device free %.1f of %.1f GiB" % (free / 2**30, total / 2**30)
            except Exception as e:
                dev = "device memory: %s" % e
            print("[mem] %5.0f s  %6d proofs  %d shapes  rss %.0f MiB  %s" % (time.time() - t0, count, len(shapes), rss, dev), flush=True)
            if os.environ.get("SOAK_GC"):
                import gc
                print("      gc.collect(): %d unreachable" % gc.collect(), flush=True)
            if os.environ.get("SOAK_MEMLOG") == "2":            # who holds it: Python allocations by source line, the C heap, the mappings
                import ctypes, gc, tracemalloc
                gc.collect()
                if tracemalloc.is_tracing():
                    snap = tracemalloc.take_snapshot()
                    for st in snap.statistics("lineno")[:6]:
                        print("      %s" % st, flush=True)
                else:
                    tracemalloc.start()
                kinds = {}
                name = None
                for line in open("/proc/self/smaps"):
                    f = line.split()
                    if "-" in f[0] and ":" not in f[0] and len(f) >= 5:
                        name = f[5] if len(f) > 5 else "[anon]"
                    elif f[0] == "Rss:":
                        kinds[name] = kinds.get(name, 0) + int(f[1])
                print("      mappings (MiB): %s" % ", ".join("%s %d" % (k, v // 1024) for k, v in sorted(kinds.items(), key=lambda kv: -kv[1])[:5]), flush=True)
    print("%d random programs in %.0f s: both prover paths byte-identical, verify() accepted every proof and rejected every altered claim; "
          "%d distinct (table heights, FRI domain) shapes, FRI domains %d..%d" % (count, time.time() - t0, len(shapes),
          min(s[-1] for s in shapes), max(s[-1] for s in shapes)))


if __name__ == "__main__":
    main()
