#!/bin/bash
# kernel + copy trace of a few 2^22-domain proofs through the production path, timeline of the last one -> gpurun_out/prove_trace_big/timeline.txt
OUT=$GRAFT_REPO_ROOT/gpurun_out/prove_trace_big
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/raw -o t -- python $GRAFT_REPO_ROOT/tools/stark_big_native.py 64 5 > $OUT/log.txt 2>&1
k=$(find $OUT/raw -name "*kernel_trace.csv" | head -1); m=$(find $OUT/raw -name "*memory_copy_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/prove_timeline.py $k $m > $OUT/timeline.txt
rm -rf $OUT/raw; tail -3 $OUT/log.txt
