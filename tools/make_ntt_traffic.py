"""profiles/ntt_traffic.json from the NTT-only PMC summary and timing that tools/prof_ntt.sh wrote (development tool).
    python tools/make_ntt_traffic.py gpurun_out/<tag> [tile_log] [round]
HBM bytes per launch = 2 * FETCH_SIZE * 1024 (gfx950: FETCH_SIZE counts 128-byte requests as 64 bytes, MI355X_MICROARCH.md "HBM") +
WRITE_SIZE * 1024, averaged over the launches of a step; valu_issue_frac = VALU wave-instructions per step * 4 cycles / (1024 SIMDs *
2.4 GHz) / step time."""
import ast
import json
import os
import re
import sys

import hashlib
import subprocess

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sources_sha16():
    """the same digest bench.py computes (kernel_sources_sha16): the figures below describe exactly these sources"""
    h = hashlib.sha256()
    for f in ("gl.hpp", "ntt_core.hpp", "ntt_plan.hpp", "ntt.hip"):
        h.update(open(os.path.join(root, "stark_brainfuck_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


d = sys.argv[1]
tl = sys.argv[2] if len(sys.argv) > 2 else "12"
rnd = sys.argv[3] if len(sys.argv) > 3 else "r04"
vals = {}
for line in open(os.path.join(d, "ntt_only_pmc_tile%s.txt" % tl)):
    m = re.match(r"(.*) dispatches (\d+) (\{.*\})", line.strip())
    if not m or "ntt_tile_kernel" not in m.group(1):
        continue
    # MODE (fifth template argument): 2 = the transposing first pass, 0 = the in-place column passes (two launches share the instantiation)
    kind = "first" if re.search(r", 2(, (true|false))?>\s*$", m.group(1).strip()) else "column"
    for k, v in ast.literal_eval(m.group(3)).items():
        vals.setdefault(k, {})[kind] = v
plain = json.loads(open(os.path.join(d, "plain_tile%s.json" % tl)).read().strip().split("\n")[-1])
col = 2 * vals["FETCH_SIZE"]["column"] * 1024 + vals["WRITE_SIZE"]["column"] * 1024
fin = 2 * vals["FETCH_SIZE"]["first"] * 1024 + vals["WRITE_SIZE"]["first"] * 1024
valu_step = 2 * vals["SQ_INSTS_VALU"]["column"] + vals["SQ_INSTS_VALU"]["first"]
ms = plain["ms_per_step"]
out = {
    "log_n": 24, "columns": 8,
    "kernel": "ntt_tile_kernel_split<4,4,0,4,MODE,NT=true> (1 transposing first-pass launch, MODE 2, + 2 in-place column launches, MODE 0, per step)",
    "source_commit": subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True, cwd=root).stdout.strip() or None,
    "kernel_sources_sha16": sources_sha16(),
    "FETCH_SIZE_KB": vals["FETCH_SIZE"], "WRITE_SIZE_KB": vals["WRITE_SIZE"],
    "correction": "gfx950: FETCH_SIZE reports 1/2 of streamed bytes (MI355X_MICROARCH.md, HBM) -> reads = 2*FETCH_SIZE*1024; WRITE_SIZE*1024 as is",
    "hbm_bytes_per_launch": (2 * col + fin) / 3, "hbm_bytes_per_launch_column": col, "hbm_bytes_per_launch_first": fin,
    "algorithmic_bytes_per_launch": 16.0 * (1 << 24) * 8 / 3,
    "SQ_INSTS_VALU_per_launch": vals["SQ_INSTS_VALU"], "valu_wave_instructions_per_step": valu_step,
    "ms_per_step_of_the_same_run": ms,
    "valu_issue_frac": valu_step * 4 / (1024 * 2.4e9) / (ms * 1e-3),
    "SQ_LDS_BANK_CONFLICT": vals.get("SQ_LDS_BANK_CONFLICT"),
    "source": "tools/prof_ntt.sh (rocprofv3 --pmc, one counter per pass, on `python tools/ntt_only.py`: the 8 x 2^24 NTT step alone); raw: profiles/%s/ntt_only_pmc.txt" % rnd,
}
import shutil
os.makedirs(os.path.join(root, "profiles", rnd), exist_ok=True)
for src, dst in (("ntt_only_pmc_tile%s.txt" % tl, "ntt_only_pmc.txt"), ("ntt_only_kernel_stats_tile%s.csv" % tl, "ntt_only_kernel_stats.csv"),
                 ("ntt_only_kernel_trace_tail_tile%s.csv" % tl, "ntt_only_kernel_trace_tail.csv"), ("plain_tile%s.json" % tl, "ntt_only_timing.json")):
    if os.path.exists(os.path.join(d, src)):
        shutil.copy(os.path.join(d, src), os.path.join(root, "profiles", rnd, dst))
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "ntt_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
