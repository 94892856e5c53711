"""Kernels around a step boundary of a rocprofv3 --kernel-trace of bench.py: the last launches of one step and the first of the next
(start offset from the boundary in us, duration, queue, name) -- what fills the idle gap between two steps."""
import csv, sys, re
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
start = sys.argv[2] if len(sys.argv) > 2 else "stem_mfma_kernel"
idx = [i for i, r in enumerate(rows) if start in r["Kernel_Name"]]
b = idx[-2]
t0 = int(rows[b]["Start_Timestamp"])
for r in rows[b - 45:b + 12]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))[:70]
    print(f"{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:8.1f} us  q{r['Queue_Id']}  {n}")
