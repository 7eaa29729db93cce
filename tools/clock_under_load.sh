#!/bin/bash
# Runs on the GPU box (via gpurun): what shader clock does the 8 x 2^24 NTT step sustain?  (i) rocm-smi / sysfs samples while
# tools/ntt_only.py loops for a few seconds, (ii) GRBM_GUI_ACTIVE (GPU-busy cycles) per dispatch next to the dispatch durations of
# the same rocprofv3 run: cycles / duration = clock.  Output -> gpurun_out/<tag>/clock.txt
set -u
TAG=${1:-clock}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python "$ROOT/tools/ntt_only.py" --steps 6000 --no-check > "$OUT/loop.json" 2>&1 &
PID=$!
sleep 4
for i in 1 2 3 4 5 6; do
  echo "--- sample $i" >> "$OUT/clock.txt"
  rocm-smi --showclocks --showpower 2>/dev/null | grep -i -E "sclk|mclk|fclk|power" >> "$OUT/clock.txt"
  for f in /sys/class/drm/card*/device/pp_dpm_sclk; do grep '\*' "$f" 2>/dev/null | sed "s|^|$f |" >> "$OUT/clock.txt"; done
  sleep 0.7
done
wait $PID
cat "$OUT/loop.json" >> "$OUT/clock.txt"
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d "$OUT/raw" -o c -- python "$ROOT/tools/ntt_only.py" --steps 10 --warmup 2 --no-check > /dev/null 2>&1
python - "$OUT" >> "$OUT/clock.txt" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
cc = glob.glob(out + "/raw/**/*counter_collection.csv", recursive=True)
kt = glob.glob(out + "/raw/**/*kernel_trace.csv", recursive=True)
dur = {}
for r in csv.DictReader(open(kt[0])):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
cyc = collections.defaultdict(float)
for r in csv.DictReader(open(cc[0])):
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        cyc[r["Dispatch_Id"]] += float(r["Counter_Value"])
rows = [(d, cyc[d], dur[d][0], dur[d][1]) for d in cyc if d in dur and "ntt_tile" in dur[d][1]]
rows = rows[len(rows) // 2:]
for d, c, ns, k in rows[:6]:
    print("dispatch", d, "GRBM_GUI_ACTIVE", c, "duration_ns", ns, "cycles/ns", round(c / ns, 3), k[:60])
tc = sum(r[1] for r in rows); tn = sum(r[2] for r in rows)
print("all ntt dispatches of the second half: cycles/ns =", round(tc / tn, 3), "(x1 if the counter is per device, /8 or /32 if it is summed over XCDs / SEs)")
PY
rm -rf "$OUT/raw"
cat "$OUT/clock.txt"
