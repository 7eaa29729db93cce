cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_merkle -o m -- python $GRAFT_REPO_ROOT/tools/merkle_only.py 22 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_merkle/**/m_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows = rows[len(rows) * 4 // 5:]     # last repetition
for r in rows:
    print(r['Kernel_Name'][:50], r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size'), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, 'us')
PY
