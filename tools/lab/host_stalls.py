"""Long HIP API calls of a rocprofv3 --hip-trace run of bench.py (host-side stalls): every call above --min us in the last part of the
trace, by name, with its start offset from the nearest preceding stem kernel launch (needs --kernel-trace in the same run)."""
import csv, sys, argparse, collections
ap = argparse.ArgumentParser()
ap.add_argument("api_csv"); ap.add_argument("--min", type=float, default=200.0); ap.add_argument("--last", type=int, default=4000)
a = ap.parse_args()
rows = sorted(csv.DictReader(open(a.api_csv)), key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-a.last:]
tot = collections.Counter(); cnt = collections.Counter()
for r in tail:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot[r["Function"]] += d; cnt[r["Function"]] += 1
    if d >= a.min:
        print(f"{int(r['Start_Timestamp']) / 1e3:14.1f} us  {d:9.1f} us  {r['Function']}")
print("-- totals over the last", len(tail), "calls")
for k, v in tot.most_common(15):
    print(f"{v / 1e3:9.2f} ms  {cnt[k]:6d}  {k}")
