"""Builds a VARIANT of libbfstark_hip.so for A/B measurements (development tool): one translation unit recompiled with extra
-D flags, linked with the product's other objects, written to tools/tmp/lib_<tag>.so (select it with BFS_LIB_PATH).

    python tools/build_variant.py <tag> <unit, e.g. ntt.hip> -DFLAG [-DFLAG ...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stark_brainfuck_amd import build as b  # noqa: E402

tag, unit, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
b.build_library()
out_dir = os.path.join(ROOT, "tools", "tmp")
os.makedirs(out_dir, exist_ok=True)
obj = os.path.join(out_dir, "%s_%s.o" % (unit, tag))
hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
compile_flags = [f for f in b.FLAGS if f != "-shared"]
subprocess.check_call([hipcc] + compile_flags + flags + ["-c", "-o", obj, os.path.join(b.CSRC, unit)], cwd=b.CSRC)
others = [os.path.join(b.OBJ, f) for f in sorted(os.listdir(b.OBJ)) if f.endswith((".hip.o", ".cpp.o")) and f != unit + ".o"]     # (not the -save-temps device objects)
lib = os.path.join(out_dir, "lib_%s.so" % tag)
subprocess.check_call([hipcc, "-shared", "-fPIC", "--offload-arch=" + b.ARCH, "-o", lib, obj] + others, cwd=b.CSRC)
print(lib)
