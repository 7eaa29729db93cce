#!/bin/bash
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_valu; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/raw -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-fri --no-check > $OUT/log.txt 2>&1
f=$(find $OUT/raw -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    agg[r["Kernel_Name"][:50]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in agg.items():
    if "ntt_tile" in k: print(k, {c: x for c, x in v.items()}, "VALU per wave", v["SQ_INSTS_VALU"] / max(v["SQ_WAVES"], 1))
PY
rm -rf $OUT/raw
