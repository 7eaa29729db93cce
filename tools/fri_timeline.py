"""Timeline of the kernels of the LAST Fri.prove in a rocprofv3 kernel trace (development tool): start offset, duration, idle gap
before each kernel.  python tools/fri_timeline.py kernel_trace.csv"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
# a prove starts with the biggest leaf kernel; find the last one
starts = [i for i, r in enumerate(rows) if "merkle_leaves_xfe_kernel" in r["Kernel_Name"] and int(r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", 0)) >= (1 << 20)]
i0 = starts[-1] if starts else 0
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = t0
busy = 0
for r in rows[i0:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"]
    name = name[:name.index("(")] if "(" in name else name
    print("%9.1f us  dur %7.1f  gap %6.1f  %s  grid %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, name[-45:], r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
    busy += e - s
    prev_end = e
print("span %.1f us, kernels %.1f us" % ((prev_end - t0) / 1e3, busy / 1e3))
