#!/bin/bash
# Round-5 evidence run on the GPU box (via gpurun): everything profiles/r05/ cites, from one box.
#   1. NTT-only kernel statistics + PMC passes (traffic, instruction classes, waits)          -> gpurun_out/r05/ntt_only_*
#   2. prover kernel statistics + PMC (Fri.prove 2^24, BrainfuckStark.prove 2^22)              -> gpurun_out/r05/{fri24,stark22}_*
#   3. per-round host timeline of Fri.prove N = 2^20 (BFS_FRI_TRACE) and its kernel timeline    -> gpurun_out/r05/fri20_*
#   4. kernel statistics of the bench command as the driver runs it                            -> gpurun_out/r05/bench_kernel_stats.csv
#   5. Hello-World proof timeline through the native stage driver                              -> gpurun_out/prove_trace_r05final/
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r05
mkdir -p "$OUT"
PMC=1 bash $ROOT/tools/prof_ntt.sh r05 > "$OUT/prof_ntt_stdout.txt" 2>&1
bash $ROOT/tools/prof_prover.sh r05 > "$OUT/prof_prover_stdout.txt" 2>&1
cd /tmp && export TMPDIR=/tmp
BFS_FRI_TRACE=1 python $ROOT/tools/fri_only.py 18 > "$OUT/fri20_host_trace.txt" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/rawfri" -o t -- python $ROOT/tools/fri_only.py 18 > "$OUT/fri20_rocprof_stdout.txt" 2>&1
k=$(find "$OUT/rawfri" -name "*kernel_trace.csv" | head -1)
python - "$k" > "$OUT/fri20_kernel_timeline.txt" <<'PY'
import csv, sys
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-44:]) for r in csv.DictReader(open(sys.argv[1])))
# the last proof: from the last merkle_leaves_xfe_kernel (round 0 hashes the input codeword) on
starts = [i for i, r in enumerate(rows) if r[2].endswith("merkle_leaves_xfe_kernel")]
i0 = starts[-1]
t0, prev, busy = rows[i0][0], rows[i0][0], 0
for s, e, name in rows[i0:]:
    print("%9.1f us  dur %7.1f  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, name))
    busy += e - s
    prev = max(prev, e)
print("span %.1f us, busy %.1f us, %d launches" % ((prev - t0) / 1e3, busy / 1e3, len(rows) - i0))
PY
find "$OUT/rawfri" -name "*kernel_stats.csv" -exec cp {} "$OUT/fri20_kernel_stats.csv" \;
rm -rf "$OUT/rawfri"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/rawbench" -o b -- python $ROOT/bench.py --steps 20 --warmup 3 > "$OUT/bench_under_rocprof.json" 2> "$OUT/bench_under_rocprof.err"
find "$OUT/rawbench" -name "*kernel_stats.csv" -exec cp {} "$OUT/bench_kernel_stats.csv" \;
rm -rf "$OUT/rawbench"
cd $ROOT
python bench.py --steps 20 --warmup 3 > "$OUT/bench_plain.json" 2> "$OUT/bench_plain.err"
bash tools/prove_trace.sh r05final > /dev/null 2>&1
ls -la "$OUT"
