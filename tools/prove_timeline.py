"""Timeline of the kernels and copies of the LAST BrainfuckStark.prove in rocprofv3 CSV traces (development tool): start offset,
duration, idle gap in front of each.   python tools/prove_timeline.py kernel_trace.csv [memcpy_trace.csv]"""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    name = name[:name.index("(")] if "(" in name else name
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name[-42:]))
if len(sys.argv) > 2:
    for r in csv.DictReader(open(sys.argv[2])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", "")) + " B"))
rows.sort()
# the last proof starts at the last occurrence of the first kernel of a proof (the randomizer's sampling kernel)
starts = [i for i, r in enumerate(rows) if "xfe_sample_kernel" in r[2]]
i0 = starts[-1] if starts else 0
t0, prev, busy = rows[i0][0], rows[i0][0], 0
for s, e, name in rows[i0:]:
    print("%9.1f us  dur %7.1f  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, name))
    busy += e - s
    prev = max(prev, e)
print("span %.1f us, busy %.1f us, %d operations" % ((prev - t0) / 1e3, busy / 1e3, len(rows) - i0))
