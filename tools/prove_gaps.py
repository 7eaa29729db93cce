"""Idle gaps between the kernels of the LAST proof in a rocprofv3 --kernel-trace database (development tool): which kernel the GPU waited
in front of, and for how long -- the host work that is not hidden behind GPU work.
    rocprofv3 --kernel-trace -d /tmp/pk -o p -- python tools/stark_prove_loop.py 64 3 && python tools/prove_gaps.py /tmp/pk/p_results.db"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = db.execute("select s.kernel_name, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start" % (kd, ks)).fetchall()
last = max(i for i, r in enumerate(rows) if "xfe_sample" in r[0])
rows = rows[last:]
gaps = []
for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
    gaps.append(((s1 - e0) / 1e3, n0.split("(")[0][-40:], n1.split("(")[0][-40:], (s1 - rows[0][1]) / 1e3))
busy = sum(e - s for _, s, e in rows) / 1e6
span = (rows[-1][2] - rows[0][1]) / 1e6
print("last proof: span %.3f ms, kernels %.3f ms, idle %.3f ms" % (span, busy, span - busy))
for g, a, b, at in sorted(gaps, reverse=True)[:18]:
    print("  idle %7.1f us at %8.1f us  after %-40s before %s" % (g, at, a, b))
