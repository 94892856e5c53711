#!/usr/bin/env python3
"""Print the top kernels of a rocprofv3 --kernel-trace --stats --output-format csv run."""
import csv
import glob
import sys

d = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 20
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
if not f:
    print("no kernel_stats.csv under", d)
    sys.exit(1)
rows = list(csv.DictReader(open(f[0])))
print(f"{'kernel':<100} {'calls':>6} {'total_ms':>10} {'avg_us':>10} {'pct':>6}")
for r in rows[:top]:
    print(f"{r['Name'][:100]:<100} {r['Calls']:>6} {float(r['TotalDurationNs'])/1e6:>10.3f} {float(r['AverageNs'])/1e3:>10.1f} {float(r['Percentage']):>6.2f}")
