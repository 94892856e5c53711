#!/usr/bin/env python3
"""Per-kernel averages of a rocprofv3 --pmc ... --output-format csv run (counter_collection.csv)."""
import csv, glob, sys, collections
d = sys.argv[1]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"    {c:32s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
