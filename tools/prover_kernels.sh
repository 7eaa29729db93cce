#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace of three 2^22-domain proofs, per-kernel breakdown of the last one -> gpurun_out/prover_kernels_2p22.txt
set -u
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/pk -o p -- python "$GRAFT_REPO_ROOT/tools/stark_prove_loop.py" 64 3 > "$GRAFT_REPO_ROOT/gpurun_out/prover_loop.txt" 2>&1
python "$GRAFT_REPO_ROOT/tools/kernel_breakdown.py" /tmp/pk/p_results.db > "$GRAFT_REPO_ROOT/gpurun_out/prover_kernels_2p22.txt" 2>&1
grep prove "$GRAFT_REPO_ROOT/gpurun_out/prover_loop.txt" | tail -1 >> "$GRAFT_REPO_ROOT/gpurun_out/prover_kernels_2p22.txt"
