"""Which cases of a tools/soak_expand.py listing the launcher sends down an expansion plan: replays the listing's (log_n, n_in)
through the planner itself (ntt_plan.hpp, compiled for the host in tests/emu: emu_expand_plan -- no arithmetic, no GPU).  The
planner does not look at the root's value beyond what the plain plan needs, so the canonical root stands in for the soak's."""
import collections, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
from build_emu import build_emulation
lib = ctypes.CDLL(build_emulation())
u32, u64 = ctypes.c_uint32, ctypes.c_uint64
lib.emu_expand_plan.argtypes = [u32, u64, u64, ctypes.POINTER(u32), ctypes.POINTER(u32)]
P = (1 << 64) - (1 << 32) + 1
taken, by_log, with_extras, total = 0, collections.Counter(), 0, 0
for line in open(sys.argv[1]):
    f = line.split()
    if len(f) != 7 or not f[0].isdigit():
        continue
    log_n, n_in = int(f[1]), int(f[2])
    m, e = u32(0), u32(0)
    total += 1
    if lib.emu_expand_plan(log_n, n_in, pow(7, (P - 1) >> log_n, P), ctypes.byref(m), ctypes.byref(e)):
        taken += 1
        by_log[log_n] += 1
        with_extras += e.value != 0
print("%d of %d cases take an expansion plan (%d of them with rank-one extras); by log2 of the domain: %s"
      % (taken, total, with_extras, " ".join("%d:%d" % kv for kv in sorted(by_log.items()))))
