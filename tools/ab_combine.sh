#!/bin/bash
# Runs on the GPU box (via gpurun): the combination kernels inside three 2^22-domain proofs, product build against variants
# (tools/build_variant.py <tag> air.hip -D...) -> gpurun_out/ab_combine.txt
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/ab_combine.txt
: > "$OUT"
for lib in "" "$@"; do
  if [ -n "$lib" ]; then export BFS_LIB_PATH=$GRAFT_REPO_ROOT/tools/tmp/lib_$lib.so; fi
  timeout 300 "$GRAFT_REPO_ROOT/tools/prover_kernels.sh"
  echo "== ${lib:-product}" >> "$OUT"
  grep -E "last proof|air_combine|zerofier|difference_combine" "$GRAFT_REPO_ROOT/gpurun_out/prover_kernels_2p22.txt" >> "$OUT"
  grep prove "$GRAFT_REPO_ROOT/gpurun_out/prover_loop.txt" | tail -1 | grep -o "'combination': [0-9.]*" >> "$OUT"
done
cat "$OUT"
