#!/bin/bash
# row_leaves_kernel (the zipped-row commitments of a 2^22-domain proof): where its issue slots go.  Separate PMC passes, kernel-trace only.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_rows
mkdir -p "$OUT"; : > "$OUT/summary.txt"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_IFETCH SQ_INSTS_FLAT SQ_ACTIVE_INST_MISC" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/raw$i" -o p -- python "$GRAFT_REPO_ROOT/tools/stark_prove_loop.py" 64 2 > "$OUT/log$i.txt" 2>&1
  f=$(find "$OUT/raw$i" -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python "$GRAFT_REPO_ROOT/tools/pmc_summary.py" "$f" | grep -E "row_leaves|air_combine_kernel<0|merkle_leaves_xfe_kernel" >> "$OUT/summary.txt"; else echo "no counter file for set $i" >> "$OUT/summary.txt"; tail -2 "$OUT/log$i.txt" >> "$OUT/summary.txt"; fi
  rm -rf "$OUT/raw$i"
done
cat "$OUT/summary.txt"
