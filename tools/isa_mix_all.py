"""Static instruction mix of the dominant NTT kernels -> profiles/r05/ntt_isa_mix.txt (development tool; no GPU needed).
The listing is made by stark_brainfuck_amd.build.build_listings() with the library's flags."""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stark_brainfuck_amd import build
lst = [p for p in build.build_listings() if os.path.basename(p).startswith("ntt-")][0]
text = open(lst).read().split("\n")
idx = [i for i, l in enumerate(text) if l.startswith("_ZN3bfs21ntt_tile_kernel_split") and ": ; @" in l]
out = ["Static instruction mix of the dominant NTT kernels: gfx950 listing made with the library's flags (stark_brainfuck_amd.build.build_listings():",
       "hipcc -O3 --offload-arch=gfx950 --cuda-device-only -S), mnemonics counted between a kernel's label and .Lfunc_end (tools/isa_mix_all.py).",
       "The 8 x 2^24 step runs ntt_tile_kernel_split<4,4,0,4,MODE,NT=true> three times: MODE 2 (PASS_FIRST: transposing, coset / padding fused) once,",
       "MODE 0 (PASS_COLUMN: in place) twice.  The code is straight-line: a thread executes the listing once per tile (16 elements x 8 butterfly levels).", ""]


def mix(start):
    counts, classes, nops = collections.Counter(), collections.Counter(), 0
    for l in text[start + 1:]:
        l = l.strip()
        if l.startswith(".Lfunc_end"):
            break
        m = re.match(r"^([a-z][a-z0-9_]+)\b", l)
        if not m or l.endswith(":"):
            continue
        op = m.group(1)
        counts[op] += 1
        cls = ("valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else
               "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other")
        if op in ("s_nop", "s_waitcnt", "s_barrier"):
            cls = op
        if op == "s_nop":
            nops += 1 + int(re.search(r"s_nop (\d+)", l).group(1))
        classes[cls] += 1
    return counts, classes, nops


for i in idx:
    name = text[i].split(":")[0]
    d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if "ntt_tile_kernel_split<4, 4, 0, 4" not in d or "true>" not in d:
        continue
    counts, classes, nops = mix(i)
    out.append(d)
    out.append("  classes: " + ", ".join("%s %d" % kv for kv in sorted(classes.items(), key=lambda kv: -kv[1])) + "; wait states spent in s_nop: %d" % nops)
    out.append("  VALU by opcode: " + ", ".join("%s %d" % kv for kv in [(k, v) for k, v in counts.most_common() if k.startswith("v_")][:24]))
    out.append("  other: " + ", ".join("%s %d" % kv for kv in [(k, v) for k, v in counts.most_common() if not k.startswith("v_")][:16]))
    mad = counts.get("v_mad_u64_u32", 0)
    out.append("  v_mad_u64_u32: %d of %d VALU (%.1f %%) (static count over all exec-masked paths; a thread EXECUTES 1 537 VALU instructions per tile, SQ_INSTS_VALU: profiles/ntt_traffic.json)"
               % (mad, classes["valu"], 100.0 * mad / classes["valu"]))
    out.append("")
os.makedirs(os.path.join(ROOT, "profiles", "r05"), exist_ok=True)
open(os.path.join(ROOT, "profiles", "r05", "ntt_isa_mix.txt"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
