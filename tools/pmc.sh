#!/bin/bash
# PMC collection for the NTT bench on the GPU box (separate passes, kernel-trace only; see MI355X_MICROARCH.md)
set -u
TAG=${1:-pmc}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/raw$i" -o p -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu --no-fri --no-check "$@" > "$OUT/log$i.txt" 2>&1
  f=$(find "$OUT/raw$i" -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
for k in agg:
    if "ntt_tile" in k:
        n = sum(1 for r in rows if r["Kernel_Name"][:60] == k) / max(1, len(agg[k]))
        print(k, "dispatches", n, {c: v / n for c, v in agg[k].items()})
PY
  else echo "no counter file for set $i"; tail -3 "$OUT/log$i.txt"; fi
  rm -rf "$OUT/raw$i"
done
