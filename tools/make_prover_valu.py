"""profiles/prover_valu.json from what tools/prof_prover.sh wrote (development tool):
    python tools/make_prover_valu.py gpurun_out/<tag> [round directory, default profiles/r04]
Per kernel of Fri.prove (N = 2^24) and of BrainfuckStark.prove (FRI domain 2^22): launches, average duration (rocprofv3 --stats), VALU
wave instructions and HBM bytes per launch (PMC; gfx950: reads = 2 * FETCH_SIZE KB, MI355X_MICROARCH.md), and the two fractions a
reader can recompute from them:  valu_issue_frac = SQ_INSTS_VALU * 4 cycles / (1024 SIMDs * clock) / duration  (clock = the GPU's
own GRBM_GUI_ACTIVE / 8 XCDs / duration when that counter was collected, else 2.4 GHz)  and  hbm_frac = bytes / duration / 8 TB/s."""
import ast
import csv
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = sys.argv[1]
rdir = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "r04")
SHORT = re.compile(r"(?:void )?(?:bfs::)?([A-Za-z_0-9]+(?:<[^>]*>)?)")


def short(name):
    name = name.strip().strip('"')
    m = SHORT.match(name)
    return m.group(1) if m else name


out = {"source": "tools/prof_prover.sh: rocprofv3 --kernel-trace --stats, and --pmc in separate passes, of `tools/fri_only.py 22` and "
                 "`tools/stark_prove_loop.py 64 3`; raw summaries copied to %s" % os.path.relpath(rdir, ROOT), "workloads": {}}
for w, label in (("fri24", "Fri.prove, N = 2^24 (codeword of 2^24 extension elements, expansion 4, 4 colinearity tests)"),
                 ("stark22", "BrainfuckStark.prove, nested-loop program of 37 254 cycles, FRI domain 2^22")):
    stats = {}
    with open(os.path.join(d, w + "_kernel_stats.csv")) as fh:
        for r in csv.DictReader(fh):
            stats[short(r["Name"])] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "total_ms": float(r["TotalDurationNs"]) / 1e6}
    pmc = {}
    for line in open(os.path.join(d, w + "_pmc.txt")):
        m = re.match(r"(.*) dispatches (\d+) (\{.*\})", line.strip())
        if m:
            pmc.setdefault(short(m.group(1)), {}).update(ast.literal_eval(m.group(3)))
    kernels = {}
    for k, s in sorted(stats.items(), key=lambda kv: -kv[1]["total_ms"])[:20]:
        # pmc names are the tail of the demangled name: match on the short name's tail
        p = next((v for n, v in pmc.items() if n == k or n.endswith(k) or k.endswith(n)), {})
        e = dict(s)
        if "SQ_INSTS_VALU" in p:
            clock = p["GRBM_GUI_ACTIVE"] / 8.0 / (s["avg_us"] * 1e-6) if p.get("GRBM_GUI_ACTIVE") else 2.4e9     # the counter is summed over the 8 XCDs
            e["valu_wave_instructions_per_launch"] = p["SQ_INSTS_VALU"]
            e["waves_per_launch"] = p.get("SQ_WAVES")
            e["clock_hz_used"] = clock
            f = p["SQ_INSTS_VALU"] * 4.0 / (1024 * clock) / (s["avg_us"] * 1e-6)
            # the instruction count, the cycle count and the duration come from three separate profiler passes (gpurun refuses combined
            # ones); for a kernel that sits on the roof their run-to-run spread can put the quotient a per cent above 1
            e["valu_issue_frac"] = min(f, 1.0)
            if f > 1.0:
                e["valu_issue_frac_uncapped"] = f
        if "FETCH_SIZE" in p and "WRITE_SIZE" in p:
            e["hbm_bytes_per_launch"] = 2 * p["FETCH_SIZE"] * 1024 + p["WRITE_SIZE"] * 1024
            e["hbm_frac"] = e["hbm_bytes_per_launch"] / (s["avg_us"] * 1e-6) / 8e12
        kernels[k] = e
    out["workloads"][w] = {"what": label, "kernels": kernels}
    for f in (w + "_kernel_stats.csv", w + "_pmc.txt", w + "_plain.txt"):
        if os.path.exists(os.path.join(d, f)):
            os.makedirs(rdir, exist_ok=True)
            shutil.copy(os.path.join(d, f), os.path.join(rdir, "prover_" + f))
json.dump(out, open(os.path.join(ROOT, "profiles", "prover_valu.json"), "w"), indent=1)
for w, v in out["workloads"].items():
    print(w)
    for k, e in v["kernels"].items():
        print("  %-44s x%-3d %9.1f us  valu %5s  hbm %5s" % (k[:44], e["calls"], e["avg_us"],
              ("%.2f" % e["valu_issue_frac"]) if "valu_issue_frac" in e else "-", ("%.3f" % e["hbm_frac"]) if "hbm_frac" in e else "-"))
