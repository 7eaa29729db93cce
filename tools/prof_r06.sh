#!/bin/bash
# Round-6 evidence run on the GPU box (via gpurun), one box for everything profiles/r06/ cites:
#   1. the full -m gpu suite                                                                   -> gpurun_out/r06/gpu_suite.txt
#   2. bench.py as the driver invokes it, then smoke()                                          -> gpurun_out/r06/bench_as_the_driver_invokes_it.json
#   3. kernel statistics of the same bench command under rocprofv3                              -> gpurun_out/r06/bench_command_kernel_stats.csv
#   4. NTT-only kernel statistics + PMC passes (traffic, VALU, LDS, waits)                      -> gpurun_out/r06/ntt_only_*
#   5. prover kernel statistics + PMC (Fri.prove 2^24, BrainfuckStark.prove 2^22)               -> gpurun_out/r06/{fri24,stark22}_*
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r06
mkdir -p "$OUT"
cd $ROOT
( time python -m pytest tests -m gpu -q -p no:cacheprovider ) > "$OUT/gpu_suite.txt" 2>&1
tail -4 "$OUT/gpu_suite.txt"
python bench.py > "$OUT/bench_as_the_driver_invokes_it.json" 2> "$OUT/bench_as_the_driver_invokes_it.err"
wc -c "$OUT/bench_as_the_driver_invokes_it.json"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$OUT/smoke.txt" 2>&1; tail -1 "$OUT/smoke.txt"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/rawbench" -o b -- python $ROOT/bench.py --steps 20 --warmup 3 > "$OUT/bench_under_rocprof.json" 2> "$OUT/bench_under_rocprof.err"
find "$OUT/rawbench" -name "*kernel_stats.csv" -exec cp {} "$OUT/bench_command_kernel_stats.csv" \;
rm -rf "$OUT/rawbench"
head -5 "$OUT/bench_command_kernel_stats.csv" | cut -c1-160
PMC=1 bash $ROOT/tools/prof_ntt.sh r06 > "$OUT/prof_ntt_stdout.txt" 2>&1
bash $ROOT/tools/prof_prover.sh r06 > "$OUT/prof_prover_stdout.txt" 2>&1
ls "$OUT"
