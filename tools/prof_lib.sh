#!/bin/bash
# kernel statistics (and with PMC="counter ...", counters) of the 8 x 2^24 NTT step for one library variant: tools/prof_lib.sh <tag> [abs lib path]
TAG=$1; LIBP=${2:-}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r06/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BFS_LIB_PATH=$LIBP rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/raw" -o ntt -- python "$ROOT/tools/ntt_only.py" --steps 30 ${NTT_ARGS:-} > "$OUT/stdout.log" 2>&1
find "$OUT/raw" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
find "$OUT/raw" -name "*kernel_trace.csv" -exec sh -c 'head -1 "$1" > "$2"; tail -12 "$1" >> "$2"' _ {} "$OUT/kernel_trace_tail.csv" \;
rm -rf "$OUT/raw"
tail -1 "$OUT/stdout.log"
cut -c1-150 "$OUT/kernel_stats.csv" | head -4
python3 - "$OUT/kernel_trace_tail.csv" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[-6:]:
    print(r.get('Kernel_Name','')[:60], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,'us')
P
: > "$OUT/pmc.txt"
for C in ${PMC:-}; do
  BFS_LIB_PATH=$LIBP rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc$C" -o ntt -- python "$ROOT/tools/ntt_only.py" --steps 3 --warmup 1 --no-check ${NTT_ARGS:-} > /dev/null 2>&1
  find "$OUT/pmc$C" -name "*counter_collection.csv" -exec python "$ROOT/tools/pmc_summary.py" {} $C \; >> "$OUT/pmc.txt"
  rm -rf "$OUT/pmc$C"
done
cat "$OUT/pmc.txt"
