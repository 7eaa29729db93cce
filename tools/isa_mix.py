"""Instruction mix of a kernel in a hipcc -S listing (development tool).
    python tools/isa_mix.py file.s <substring of the mangled kernel name> [--top N]"""
import collections
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 25
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and ":" in l and not l.startswith("\t"))
    counts, classes = collections.Counter(), collections.Counter()
    for l in lines[start + 1:]:
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
        classes[cls] += 1
    print(lines[start])
    print(dict(classes))
    nops = sum(1 + int(re.search(r"s_nop (\d+)", l).group(1)) for l in lines[start + 1:] if re.search(r"^\s*s_nop (\d+)", l) and True)
    print("top:", counts.most_common(top))


main()
