"""Per-kernel average of one rocprofv3 counter_collection.csv (development tool): python tools/pmc_summary.py file.csv [label]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in rows:
    k = r["Kernel_Name"]
    k = k[:k.index("(")] if "(" in k else k
    k = k[-70:]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[k].add(r.get("Dispatch_Id", r.get("Correlation_Id", "")))
for k in sorted(agg):
    n = max(1, len(disp[k]))
    print(k, "dispatches", n, {c: round(v / n, 1) for c, v in agg[k].items()})
