"""Builds a variant of libbfstark_hip.so whose DEVICE code of one translation unit went through an assembly-level edit (development
tool for timing-only experiments: hazards and results are the experimenter's problem).

    python tools/build_from_asm.py <tag> <unit, e.g. ntt.hip> <edit> [-DFLAG ...]      -> tools/tmp/lib_<tag>.so (BFS_LIB_PATH)

edits:  none         reassemble unchanged (checks the pipeline)
        strip_snop   delete every s_nop (the wait states between a VALU write of VCC / an SGPR and its VALU read, which hipcc inserts
                     after every carry-producing instruction on gfx950): an upper bound on what explicit SGPR-pair carry chains could
                     buy, without writing them

The steps are hipcc's own (`hipcc -###`): device cc1 (here stopped at -S), assembler, lld, clang-offload-bundler, host cc1."""
import os
import re
import shlex
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stark_brainfuck_amd import build as b  # noqa: E402

tag, unit, edit, flags = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4:]
b.build_library()
tmp = os.path.join(ROOT, "tools", "tmp")
os.makedirs(tmp, exist_ok=True)
hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
src = os.path.join(b.CSRC, unit)
obj = os.path.join(tmp, "%s_%s.o" % (unit, tag))
compile_flags = [f for f in b.FLAGS if f != "-shared"]
plan = subprocess.run([hipcc, "-###"] + compile_flags + flags + ["-c", "-o", obj, src], cwd=b.CSRC, capture_output=True, text=True).stderr
cmds = [shlex.split(l) for l in plan.splitlines() if l.startswith(' "')]
dev = next(c for c in cmds if "-cc1" in c and "amdgcn-amd-amdhsa" in c[c.index("-triple") + 1])
lld = next(c for c in cmds if c[0].endswith("lld"))
bundler = next(c for c in cmds if c[0].endswith("clang-offload-bundler"))
host = next(c for c in cmds if "-cc1" in c and c[c.index("-triple") + 1].startswith("x86_64"))
dev_obj = dev[dev.index("-o") + 1]
asm = os.path.join(tmp, "%s_%s.s" % (unit, tag))
dev_s = [asm if a == dev_obj else ("-S" if a == "-emit-obj" else a) for a in dev]
subprocess.check_call(dev_s, cwd=b.CSRC)
text = open(asm).read()
if edit == "strip_snop":
    text, n = re.subn(r"^\s*s_nop\s+\d+\s*\n", "", text, flags=re.M)
    print("removed %d s_nop" % n)
elif edit != "none":
    raise SystemExit("unknown edit " + edit)
open(asm, "w").write(text)
clang = dev[0]
subprocess.check_call([clang, "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", asm, "-o", dev_obj])
for c in (lld, bundler, host):
    subprocess.check_call(c, cwd=b.CSRC)
others = [os.path.join(b.OBJ, f) for f in sorted(os.listdir(b.OBJ)) if f.endswith(".o") and f != unit + ".o"]
lib = os.path.join(tmp, "lib_%s.so" % tag)
subprocess.check_call([hipcc, "-shared", "-fPIC", "--offload-arch=" + b.ARCH, "-o", lib, obj] + others, cwd=b.CSRC)
print(lib)
