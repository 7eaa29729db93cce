"""Reads a rocprofv3 --kernel-trace --hip-trace --memory-copy-trace database and prints the long events of the last third of the run, plus
everything around stream synchronisations longer than 15 ms (how the dispatch stalls after pinning / unpinning host memory were found).
"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
print([t.split("_0000")[0] if "_0000" in t else t for t in tabs][:40])
def tab(key):
    c = [t for t in tabs if key in t]
    return c[0] if c else None
kd, ks = tab("kernel_dispatch"), tab("kernel_symbol")
rows = db.execute("select s.kernel_name, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start" % (kd, ks)).fetchall()
t0 = rows[0][1]
api = tab("rocpd_region")
print("region cols", [r[1] for r in db.execute("pragma table_info(%s)" % api)])
st = tab("rocpd_string")
regs = db.execute("select s.string, r.start, r.end from %s r join %s s on r.name_id = s.id order by r.start" % (api, st)).fetchall()
print(len(rows), "kernels", len(regs), "regions")
mc = tab("memory_copy")
copies = []
if mc:
    cols = [r[1] for r in db.execute("pragma table_info(%s)" % mc)]
    print("copy cols", cols)
    copies = db.execute("select start, end, size from %s order by start" % mc).fetchall()
# window: the last third of the run
lo = rows[len(rows) * 2 // 3][1]
events = []
for name, s, e in rows:
    if s >= lo and (e - s) > 200000: events.append((s, e, "K " + name[:60]))
for name, s, e in regs:
    if s >= lo and (e - s) > 300000: events.append((s, e, "A " + name[:60]))
for s, e, size in copies:
    if s >= lo and ((e - s) > 300000 or size > (1 << 20)): events.append((s, e, "C %d bytes" % size))
events.sort()
for s, e, what in events:
    print("%10.3f -> %10.3f  (%8.3f ms)  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, what))
print("=========== detail around long synchronisations")
longs = [(s, e) for name, s, e in regs if name == "hipStreamSynchronize" and e - s > 15e6 and s >= lo]
for (ls, le) in longs[:2]:
    w0, w1 = ls - 8e6, le + 1e6
    ev = []
    for name, s, e in rows:
        if w0 <= s <= w1: ev.append((s, e, "K " + name[:70]))
    for name, s, e in regs:
        if w0 <= s <= w1: ev.append((s, e, "A " + name[:70]))
    for s, e, size in copies:
        if w0 <= s <= w1: ev.append((s, e, "C %d bytes" % size))
    ev.sort()
    print("---- window", (w0 - t0) / 1e6, (w1 - t0) / 1e6, len(ev), "events")
    # compress runs of identical names
    last, cnt = None, 0
    for s, e, what in ev:
        if (e - s) < 50000 and what == last:
            cnt += 1
            continue
        if cnt: print("            ... %d more %s" % (cnt, last))
        cnt = 0
        last = what
        print("%10.3f -> %10.3f  (%8.3f ms)  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, what))
    if cnt: print("            ... %d more %s" % (cnt, last))
