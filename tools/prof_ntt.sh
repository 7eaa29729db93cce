#!/bin/bash
# Runs on the GPU box (via gpurun): NTT-only rocprofv3 kernel statistics (and, with PMC=1, HBM traffic counters in their own
# passes) of the 8 x 2^24 step for each tile size.  Summaries -> gpurun_out/<tag>/ ; copy what should be judged to profiles/.
# usage: tools/prof_ntt.sh <tag> [tile logs...]
set -u
TAG=${1:-ntt}; shift || true
TILES=${@:-12}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for TL in $TILES; do
  : # (one tile size since round 2; the loop variable only names the output files)
  BFS_NTT_WS_PROBE_LOG=1 python "$ROOT/tools/ntt_only.py" --steps 30 > "$OUT/plain_tile$TL.json" 2> "$OUT/route_probe_tile$TL.log"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/raw$TL" -o ntt -- python "$ROOT/tools/ntt_only.py" --steps 30 > "$OUT/rocprof_stdout_tile$TL.log" 2>&1
  find "$OUT/raw$TL" -name "*kernel_stats.csv" -exec cp {} "$OUT/ntt_only_kernel_stats_tile$TL.csv" \;
  find "$OUT/raw$TL" -name "*kernel_trace.csv" -exec sh -c 'head -1 "$1" > "$2"; tail -90 "$1" >> "$2"' _ {} "$OUT/ntt_only_kernel_trace_tail_tile$TL.csv" \;
  rm -rf "$OUT/raw$TL"
  if [ "${PMC:-0}" = "1" ]; then
    for C in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY; do
      rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc$TL$C" -o ntt -- python "$ROOT/tools/ntt_only.py" --steps 3 --warmup 1 --no-check > /dev/null 2>&1
      find "$OUT/pmc$TL$C" -name "*counter_collection.csv" -exec python "$ROOT/tools/pmc_summary.py" {} $C \; >> "$OUT/ntt_only_pmc_tile$TL.txt"
      rm -rf "$OUT/pmc$TL$C"
    done
  fi
  cat "$OUT/plain_tile$TL.json"; grep "ntt route" "$OUT/route_probe_tile$TL.log"
  head -8 "$OUT/ntt_only_kernel_stats_tile$TL.csv"
done
