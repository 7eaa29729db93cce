"""Per-kernel time of the LAST proof in a rocprofv3 --kernel-trace database (sqlite):
    rocprofv3 --kernel-trace -d /tmp/pk -o p -- python tools/stark_prove_loop.py 64 3 && python tools/kernel_breakdown.py /tmp/pk/p_results.db
"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = db.execute("select s.kernel_name, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start" % (kd, ks)).fetchall()
# last proof = after the last long gap... simply take kernels after the last occurrence of xfe_sample_kernel
last = max(i for i, r in enumerate(rows) if "xfe_sample" in r[0])
agg = collections.OrderedDict()
for name, s, e in rows[last:]:
    k = name.split("(")[0][:70]
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1; a[1] += (e - s) / 1e6
tot = sum(v[1] for v in agg.values())
print("last proof: %d kernels, %.3f ms of kernel time, span %.3f ms" % (len(rows) - last, tot, (rows[-1][2] - rows[last][1]) / 1e6))
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print("%8.3f ms %5d  %s" % (t, c, k))
