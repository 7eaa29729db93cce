#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/fri_trace_${1:-x}
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $OUT/raw -o t -- python $GRAFT_REPO_ROOT/tools/fri_only.py > $OUT/log.txt 2>&1
find $OUT/raw -name "*kernel_trace.csv" -exec cp {} $OUT/kernel_trace.csv \;
find $OUT/raw -name "*memory_copy_trace.csv" -exec cp {} $OUT/memcpy_trace.csv \;
find $OUT/raw -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/raw; grep "ms" $OUT/log.txt | tail -2
