#!/bin/bash
# A/B of NTT variants on the GPU box: for each "VAR=VAL,VAR=VAL" spec, the plain timing line and per-pass steady-state kernel
# durations from a rocprofv3 kernel trace.   usage: tools/ab_ntt.sh <tag> spec...
TAG=$1; shift
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for SPEC in "$@"; do
  NAME=$(echo "$SPEC" | tr ',=/' '___')
  ( IFS=,; for kv in $SPEC; do export "$kv"; done
    python "$ROOT/tools/ntt_only.py" --steps 50 > "$OUT/plain_$NAME.json" 2>&1
    rocprofv3 --kernel-trace --output-format csv -d "$OUT/raw_$NAME" -o ntt -- python "$ROOT/tools/ntt_only.py" --steps 40 --no-check > /dev/null 2>&1
    f=$(find "$OUT/raw_$NAME" -name "*kernel_trace.csv" | head -1)
    echo "== $SPEC"; tail -1 "$OUT/plain_$NAME.json"
    python "$ROOT/tools/per_pass.py" "$f" | tee "$OUT/per_pass_$NAME.txt"
    rm -rf "$OUT/raw_$NAME"
    if [ "${PMC:-0}" = "1" ]; then       # VALU wave instructions per launch, in its own pass (kernel-trace only)
      rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU --output-format csv -d "$OUT/pmc_$NAME" -o ntt -- python "$ROOT/tools/ntt_only.py" --steps 3 --warmup 1 --no-check > /dev/null 2>&1
      find "$OUT/pmc_$NAME" -name "*counter_collection.csv" -exec python "$ROOT/tools/pmc_summary.py" {} \; | grep ntt_tile | sed 's/^/  /'
      rm -rf "$OUT/pmc_$NAME"
    fi )
done
