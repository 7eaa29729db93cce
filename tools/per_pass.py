"""Steady-state duration of each NTT pass from a rocprofv3 kernel trace (development tool): the last 30 steps' dispatches of
ntt_tile_kernel, grouped by position inside a step."""
import csv
import statistics
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "ntt_tile_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
npass = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = rows[-30 * npass:]
total = 0.0
for k in range(npass):
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[k::npass]]
    name = rows[k]["Kernel_Name"]
    name = name[name.index("<"):name.index(">") + 1]
    total += statistics.median(d)
    print("  pass %d %-18s median %7.1f us  min %7.1f  max %7.1f" % (k, name, statistics.median(d) / 1e3, min(d) / 1e3, max(d) / 1e3))
span = [int(rows[i + npass]["Start_Timestamp"]) - int(rows[i]["Start_Timestamp"]) for i in range(0, len(rows) - npass, npass)]
print("  sum of medians %.1f us; step-to-step %.1f us" % (total / 1e3, statistics.median(span) / 1e3))
