#!/usr/bin/env python3
"""Idle gaps of the last full step in a rocprofv3 kernel trace of bench.py: where the whole device waits, and between which kernels."""
import csv, re, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "stem_mfma_kernel" in r["Kernel_Name"]]
seg = rows[starts[-2]:starts[-1]]
t0 = int(seg[0]["Start_Timestamp"])
short = lambda n: re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("void ", "").replace("mt::", ""))[:44]
end, gaps = 0, []
for i, r in enumerate(seg):
    s = int(r["Start_Timestamp"])
    if end and s - end > 4000:
        gaps.append((s - end, (end - t0) / 1e6, short(seg[i - 1]["Kernel_Name"]) + " q" + seg[i - 1]["Queue_Id"], short(r["Kernel_Name"]) + " q" + r["Queue_Id"]))
    end = max(end, int(r["End_Timestamp"]))
print(f"step {(end - t0) / 1e6:.2f} ms; {len(gaps)} gaps > 4 us totalling {sum(g[0] for g in gaps) / 1e6:.2f} ms")
by = collections.defaultdict(lambda: [0, 0])
for g in gaps:
    k = g[2].split("<")[0] + " -> " + g[3].split("<")[0]
    by[k][0] += 1; by[k][1] += g[0]
for k, (n, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"  {t / 1e3:7.1f} us in {n:3d} gaps  {k}")
