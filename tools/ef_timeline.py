#!/usr/bin/env python3
"""Prints the last step of a tools/ef_trace.sh kernel trace: every launch >= --min us, plus per-kernel-family totals."""
import argparse
import csv
import re
from collections import defaultdict

ap = argparse.ArgumentParser()
ap.add_argument("csv")
ap.add_argument("--min", type=float, default=40.0)
ap.add_argument("--families", action="store_true")
a = ap.parse_args()
rows = sorted(csv.DictReader(open(a.csv)), key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "").replace("mt::", "")
    return re.sub(r"\(.*", "", n)[:56]


names = [short(r["Kernel_Name"]) for r in rows]
stem = [i for i, n in enumerate(names) if "stem_mfma_kernel" in n]               # the stem kernel starts a step
seg = rows[stem[-1]:]
t0 = int(seg[0]["Start_Timestamp"])
fam = defaultdict(lambda: [0, 0.0])
tot = 0.0
for i, r in enumerate(seg):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    f = fam[re.sub(r"<.*", "", short(r["Kernel_Name"]))]
    f[0] += 1
    f[1] += d
    if d >= a.min and not a.families:
        print(f"{i:4d} {(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {d:8.1f}  {short(r['Kernel_Name'])} g={r['Grid_Size_X']}")
print(f"kernel sum {tot / 1e3:.2f} ms; span {(int(seg[-1]['End_Timestamp']) - t0) / 1e6:.2f} ms; {len(seg)} launches")
for k, (n, d) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print(f"  {d / 1e3:7.2f} ms {n:4d}  {k}")
