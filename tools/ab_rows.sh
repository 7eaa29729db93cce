#!/bin/bash
# Runs on the GPU box (via gpurun): the zipped-row leaf kernel inside three 2^22-domain proofs, product build against timing-only
# variants (tools/build_variant.py <tag> rows.hip -D...) -> gpurun_out/ab_rows.txt
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/ab_rows.txt
: > "$OUT"
for lib in "" "$@"; do
  if [ -n "$lib" ]; then export BFS_LIB_PATH=$GRAFT_REPO_ROOT/tools/tmp/lib_$lib.so; fi
  timeout 300 "$GRAFT_REPO_ROOT/tools/prover_kernels.sh"
  echo "== ${lib:-product}" >> "$OUT"
  grep -E "last proof|row_leaves" "$GRAFT_REPO_ROOT/gpurun_out/prover_kernels_2p22.txt" >> "$OUT"
done
cat "$OUT"
