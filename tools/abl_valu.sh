#!/bin/bash
cd /tmp && export TMPDIR=/tmp
for f in $GRAFT_REPO_ROOT/gpurun_abl_*.so; do
OUT=/tmp/pv; rm -rf $OUT; mkdir -p $OUT
BFS_LIB_PATH=$f rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/raw -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-fri --no-check > $OUT/log.txt 2>&1
c=$(find $OUT/raw -name "*counter_collection.csv" | head -1)
python3 - "$c" "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(float)
for r in rows:
    if "ntt_tile" in r["Kernel_Name"]: agg[r["Counter_Name"]] += float(r["Counter_Value"])
print(sys.argv[2].split("/")[-1], "VALU per wave %.0f" % (agg["SQ_INSTS_VALU"] / max(agg["SQ_WAVES"], 1)))
PY
done
