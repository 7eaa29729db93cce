"""Does a loop of proofs grow the process?  (development tool)  Resident set and the library's pool figures per 100 proofs, for the
production path, the Python stages with intermediates kept, and verify(), each with the operating system's randomness and with a
replaced urandom (explicit salts, as the soaks and golden tests use).  usage: python tools/leak_probe.py [proofs per leg]"""
import ctypes, gc, hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stark_brainfuck_amd import _lib, brainfuck_stark, salted_merkle, table
from stark_brainfuck_amd.brainfuck_stark import BrainfuckStark
from stark_brainfuck_amd.vm import VirtualMachine

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
code = os.environ.get("PROBE_CODE", "++++[>++++[>++<-]<-]>>.")
program = VirtualMachine.compile(code)
rt, inp, out = VirtualMachine.run(program)
m = VirtualMachine.simulate(program, input_data=inp)


class Stream:
    def __init__(self, tag):
        self.tag, self.pos, self.buf = tag, 0, b""

    def __call__(self, n):
        end = self.pos + n
        if end > len(self.buf):
            self.buf = hashlib.shake_256(b"probe" + self.tag).digest(max(2 * end, 1 << 16))
        o = self.buf[self.pos:end]
        self.pos = end
        return o


def rss():
    return int(open("/proc/self/statm").read().split()[1]) * os.sysconf("SC_PAGE_SIZE") / 2**20


class _Mallinfo2(ctypes.Structure):
    _fields_ = [(n, ctypes.c_size_t) for n in ("arena", "ordblks", "smblks", "hblks", "hblkhd", "usmblks", "fsmblks", "uordblks", "fordblks", "keepcost")]


def heap():
    """bytes the C allocator has handed out (malloc arenas + mmapped blocks), MiB"""
    try:
        libc = ctypes.CDLL("libc.so.6")
        libc.mallinfo2.restype = _Mallinfo2
        mi = libc.mallinfo2()
        return (mi.uordblks + mi.hblkhd) / 2**20
    except Exception:
        return float("nan")


def smaps():
    """resident KiB per mapping name (anonymous mappings grouped by size class of the mapping)"""
    out, name, size = {}, None, 0
    for line in open("/proc/self/smaps"):
        f = line.split()
        if "-" in f[0] and len(f) >= 5 and ":" not in f[0]:
            name = f[5] if len(f) > 5 else "[anon]"
        elif f[0] == "Size:":
            size = int(f[1])
            if name == "[anon]":
                name = "[anon %s]" % ("< 1 MiB" if size < 1024 else "< 64 MiB" if size < 65536 else ">= 64 MiB")
        elif f[0] == "Rss:":
            out[name] = out.get(name, 0) + int(f[1])
    return out


def pools():
    lib = _lib.load()
    live, cached = ctypes.c_size_t(), ctypes.c_size_t()
    lib.bfs_pool_stats(ctypes.byref(live), ctypes.byref(cached))
    return live.value / 2**20, cached.value / 2**20


def leg(name, fn):
    import tracemalloc
    fn()
    gc.collect()
    a, ha, maps = rss(), heap(), smaps()
    tracemalloc.start()
    before = tracemalloc.take_snapshot()
    t = time.time()
    for k in range(reps):
        fn()
    gc.collect()
    after = tracemalloc.take_snapshot()
    tracemalloc.stop()
    py = sum(st.size_diff for st in after.compare_to(before, "filename")) / 2**20
    print("%-52s %4d calls  rss %7.0f -> %7.0f MiB  (%+.3f MiB per call; C heap in use %+.3f, Python heap %+.3f per call)  device pool live %.0f cached %.0f MiB  %.1f s"
          % (name, reps, a, rss(), (rss() - a) / reps, (heap() - ha) / reps, py / reps, pools()[0], pools()[1], time.time() - t), flush=True)
    try:
        libc = ctypes.CDLL("libc.so.6")
        libc.mallinfo2.restype = _Mallinfo2
        mi = libc.mallinfo2()
        before_trim = rss()
        libc.malloc_trim(0)
        print("      main arena: %.0f MiB in use, %.0f MiB free, %.0f MiB mmapped; malloc_trim(0): rss %.0f -> %.0f MiB"
              % (mi.uordblks / 2**20, mi.fordblks / 2**20, mi.hblkhd / 2**20, before_trim, rss()))
    except Exception as e:
        print("      mallinfo2 / malloc_trim: %s" % e)
    now = smaps()
    grown = sorted(((now.get(k, 0) - maps.get(k, 0), k) for k in now), reverse=True)[:4]
    print("      mappings that grew (KiB): %s" % ", ".join("%s %+d" % (k, d) for d, k in grown if d > 256))
    top = sorted(after.compare_to(before, "lineno"), key=lambda st: -st.size_diff)[:3]
    for st in top:
        if st.size_diff > 64 * 1024:
            print("      %s" % st)


original = (brainfuck_stark.urandom, salted_merkle.urandom, table.urandom)
counter = [0]


def prove(keep, replaced):
    def run():
        if replaced:
            counter[0] += 1
            s = Stream(str(counter[0]).encode())
            brainfuck_stark.urandom = salted_merkle.urandom = table.urandom = s
        else:
            brainfuck_stark.urandom, salted_merkle.urandom, table.urandom = original
        stark = BrainfuckStark(rt, len(m[1]), program, inp, out)
        stark.keep_intermediates = keep
        run.proof = stark.prove(program, *m)
    return run


print("program %r: %d cycles, FRI domain %d" % (code, rt, BrainfuckStark(rt, len(m[1]), program, inp, out).fri.domain.length))
p = prove(False, False)
leg("production path, os.urandom", p)
leg("production path, replaced urandom (explicit salts)", prove(False, True))
if os.environ.get("PROBE_PYTHON_STAGES", "1") == "1":
    leg("Python stages + intermediates, os.urandom", prove(True, False))
    leg("Python stages + intermediates, replaced urandom", prove(True, True))
proof = p.proof
leg("verify()", lambda: BrainfuckStark(rt, len(m[1]), program, inp, out).verify(proof))
# the hot-path entry points through the reference's own call surface (the Python mirror), collector off: nothing may wait for it
if os.environ.get("PROBE_API", "1") == "1":
    import stark_brainfuck_amd as sb
    from stark_brainfuck_amd.algebra import BaseField, BaseFieldElement
    from stark_brainfuck_amd.extension_field import ExtensionField
    from stark_brainfuck_amd.fri import Fri
    from stark_brainfuck_amd.ip import ProofStream
    from stark_brainfuck_amd.merkle import Merkle
    from stark_brainfuck_amd.ntt import fast_coset_evaluate, ntt
    from stark_brainfuck_amd.univariate import Polynomial
    field, xfield = BaseField.main(), ExtensionField.main()
    nn = 1 << 12
    omega = field.primitive_nth_root(nn)
    values = [BaseFieldElement((i * i + 7) % field.p, field) for i in range(nn)]
    poly = Polynomial(values[:nn // 4])
    fri = Fri(field.generator(), omega, nn, 4, 2, xfield)
    xcoeffs = [xfield.sample(bytes([i % 251, 1, 2, 3] * 8)) for i in range(nn // 4)]
    codeword = fri.domain.xevaluate(Polynomial(xcoeffs))
    gc.disable()
    leg("ntt(2^12) through the mirror", lambda: ntt(omega, values))
    leg("fast_coset_evaluate(2^10 -> 2^12)", lambda: fast_coset_evaluate(poly, field.generator(), omega, nn))
    leg("Merkle(2^12 extension leaves) + root + open", lambda: Merkle(codeword).open(5))
    leg("Fri.prove(2^12 extension codeword)", lambda: fri.prove(codeword, ProofStream()))
    gc.enable()
gc.collect()
print("after gc.collect(): rss %.0f MiB" % rss())
