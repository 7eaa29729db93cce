#!/bin/bash
# PMC counters of the prover's kernels at FRI domain 2^22 (separate passes, kernel-trace only; see MI355X_MICROARCH.md) -> gpurun_out/pmc_prover.txt
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_prover
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
: > "$GRAFT_REPO_ROOT/gpurun_out/pmc_prover.txt"
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/raw$i" -o p -- python "$GRAFT_REPO_ROOT/tools/stark_prove_loop.py" 64 2 > "$OUT/log$i.txt" 2>&1
  f=$(find "$OUT/raw$i" -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 - "$f" >> "$GRAFT_REPO_ROOT/gpurun_out/pmc_prover.txt" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in rows:
    k = r["Kernel_Name"].split("(")[0][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[k].add(r["Dispatch_Id"])
for k in sorted(agg):
    if any(s in k for s in ("row_leaves", "air_combine_kernel<0>", "air_combine_kernel<1>", "merkle_leaves_xfe_kernel", "zerofier", "merkle_parents_kernel")):
        n = len(disp[k])
        print(k, "dispatches", n, {c: round(v / n, 1) for c, v in sorted(agg[k].items())})
PY
  else echo "no counter file for set $i" >> "$GRAFT_REPO_ROOT/gpurun_out/pmc_prover.txt"; tail -3 "$OUT/log$i.txt" >> "$GRAFT_REPO_ROOT/gpurun_out/pmc_prover.txt"; fi
  rm -rf "$OUT/raw$i"
done
