#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + stats of the default bench, summaries -> gpurun_out/prof_<tag>/
# usage: tools/profile_bench.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/raw" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 2 --no-cpu "$@" > "$OUT/bench_stdout.log" 2>&1
find "$OUT/raw" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
find "$OUT/raw" -name "*kernel_trace.csv" -exec sh -c 'head -400 "$1" > "$2"' _ {} "$OUT/kernel_trace_head.csv" \;
rm -rf "$OUT/raw"
tail -2 "$OUT/bench_stdout.log"
cat "$OUT/kernel_stats.csv" | head -30
