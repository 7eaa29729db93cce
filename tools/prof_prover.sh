#!/bin/bash
# Runs on the GPU box (via gpurun): per-kernel evidence for the kernels that dominate Fri.prove (N = 2^24) and BrainfuckStark.prove
# (FRI domain 2^22) -- the counterpart of tools/prof_ntt.sh for everything that is not the NTT.
#   1. rocprofv3 --kernel-trace --stats of each command        -> <tag>/{fri24,stark22}_kernel_stats.csv
#   2. PMC in separate passes (kernel-trace only, one set each) -> <tag>/{fri24,stark22}_pmc.txt  (per kernel, averaged per dispatch)
# then `python tools/make_prover_valu.py gpurun_out/<tag>` turns both into profiles/prover_valu.json (what bench.py reads).
# usage: tools/prof_prover.sh <tag>
set -u
TAG=${1:-prover}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
declare -A CMD
CMD[fri24]="python $ROOT/tools/fri_only.py 22"
CMD[stark22]="python $ROOT/tools/stark_prove_loop.py 64 3"
for W in fri24 stark22; do
  ${CMD[$W]} > "$OUT/${W}_plain.txt" 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/raw_$W" -o p -- ${CMD[$W]} > "$OUT/${W}_rocprof_stdout.txt" 2>&1
  find "$OUT/raw_$W" -name "*kernel_stats.csv" -exec cp {} "$OUT/${W}_kernel_stats.csv" \;
  rm -rf "$OUT/raw_$W"
  : > "$OUT/${W}_pmc.txt"
  for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
    rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$OUT/pmc_$W" -o p -- ${CMD[$W]} > /dev/null 2>&1
    f=$(find "$OUT/pmc_$W" -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then python "$ROOT/tools/pmc_summary.py" "$f" >> "$OUT/${W}_pmc.txt"; else echo "no counter file for: $SET" >> "$OUT/${W}_pmc.txt"; fi
    rm -rf "$OUT/pmc_$W"
  done
done
head -12 "$OUT/fri24_kernel_stats.csv" "$OUT/stark22_kernel_stats.csv"
