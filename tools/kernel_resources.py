#!/usr/bin/env python3
"""Register / LDS / occupancy table of every kernel the library ships, straight from the compiler
(hipcc -Rpass-analysis=kernel-resource-usage on each translation unit with the flags of stark_brainfuck_amd/build.py).
Runs without a GPU.  The output is tracked per round so that a register regression shows up in a diff:

    python tools/kernel_resources.py > profiles/r03/kernel_resources.txt
    python tools/kernel_resources.py air.hip -DSOME_VARIANT          (one file, extra flags: for A/B work)
"""
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stark_brainfuck_amd import build as b  # noqa: E402


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        return [re.sub(r"\bbfs::", "", o) for o in out[:len(names)]]
    except OSError:
        return names


def resources(src, extra):
    flags = [f for f in b.FLAGS if f != "-shared"] + list(extra)
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + flags + ["-c", "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage", src]
    res = subprocess.run(cmd, cwd=b.CSRC, capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stderr)
        raise SystemExit("hipcc failed on " + src)
    rows, cur = [], None
    for line in res.stderr.splitlines():
        m = re.search(r"remark: +(Function Name|[A-Za-z ]+(?: \[[^\]]+\])?): (.+?) \[-Rpass", line)
        if not m:
            continue
        key, val = m.group(1).strip(), m.group(2).strip()
        if key == "Function Name":
            cur = {"name": val}
            rows.append(cur)
        elif cur is not None:
            cur[key] = val
    return rows


def main():
    args = sys.argv[1:]
    files = [a for a in args if a.endswith(".hip")]
    extra = [a for a in args if not a.endswith(".hip")]
    srcs = [os.path.join(b.CSRC, f) for f in files] if files else [s for s in b.sources() if s.endswith(".hip")]
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        tables = list(pool.map(lambda s: resources(s, extra), srcs))
    print("# kernel resources, gfx950, flags: %s" % " ".join([f for f in b.FLAGS if f != "-shared"] + extra))
    print("# %-26s %5s %5s %5s %8s %5s %8s  %s" % ("file", "VGPR", "AGPR", "SGPR", "scratch", "occ", "LDS", "kernel"))
    for src, rows in zip(srcs, tables):
        names = demangle([r["name"] for r in rows])
        for r, nm in sorted(zip(rows, names), key=lambda t: t[1]):
            print("%-28s %5s %5s %5s %8s %5s %8s  %s" % (os.path.basename(src), r.get("VGPRs", "?"), r.get("AGPRs", "?"), r.get("SGPRs", r.get("TotalSGPRs", "?")),
                                                        r.get("ScratchSize [bytes/lane]", "?"), r.get("Occupancy [waves/SIMD]", "?"),
                                                        r.get("LDS Size [bytes/block]", "?"), nm))


if __name__ == "__main__":
    main()
