#!/bin/bash
# Runs on the GPU box (via gpurun): Fri.prove at N = 2^24 and 2^20 and the leaf kernels' statistics, product build against variants
# (tools/build_variant.py <tag> merkle.hip -D...) -> gpurun_out/ab_leaf.txt
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/ab_leaf.txt
: > "$OUT"
cd /tmp && export TMPDIR=/tmp
for lib in "" "$@" ""; do
  if [ -n "$lib" ]; then export BFS_LIB_PATH=$ROOT/tools/tmp/lib_$lib.so; else unset BFS_LIB_PATH; fi
  echo "== ${lib:-product}" >> "$OUT"
  python "$ROOT/tools/fri_only.py" 22 | python -c "import sys,ast; d=ast.literal_eval(sys.stdin.read().strip().splitlines()[-1]); print('  Fri.prove 2^24: %.3f ms, rounds %.3f' % (d['ms'], d['breakdown_ms']['rounds']))" >> "$OUT"
  python "$ROOT/tools/fri_only.py" 18 | python -c "import sys,ast; d=ast.literal_eval(sys.stdin.read().strip().splitlines()[-1]); print('  Fri.prove 2^20: %.3f ms, rounds %.3f' % (d['ms'], d['breakdown_ms']['rounds']))" >> "$OUT"
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl -o p -- python "$ROOT/tools/fri_only.py" 22 > /dev/null 2>&1
  f=$(find /tmp/abl -name "*kernel_stats.csv" | head -1)
  python - "$f" >> "$OUT" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r["Name"] for k in ("merkle_leaves_xfe", "merkle_parents_kernel")):
        print("  %-34s calls %4s  total %8.3f ms  avg %8.1f us" % (r["Name"].split("(")[0][-34:], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
  rm -rf /tmp/abl
done
cat "$OUT"
