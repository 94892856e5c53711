#!/usr/bin/env python3
"""Device-idle gaps of the last full step of a rocprofv3 trace (kernel + memory-copy + HIP runtime): for every gap > --min us with no
kernel in flight on any queue, the kernels / copies on either side and the HIP runtime calls that were in progress on the host."""
import argparse, csv, glob, re

ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("--min", type=float, default=20.0)
a = ap.parse_args()


def load(pat):
    f = glob.glob(f"{a.dir}/**/*{pat}.csv", recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "").replace("mt::", "")
    return re.sub(r"\(.*", "", n)[:60]


kt = sorted(load("kernel_trace"), key=lambda r: int(r["Start_Timestamp"]))
mc = load("memory_copy_trace")
api = load("hip_api_trace")
starts = [i for i, r in enumerate(kt) if "stem_mfma_kernel" in r["Kernel_Name"]]
lo, hi = starts[-2], starts[-1]
seg = kt[lo:hi + 1]
t0 = int(seg[0]["Start_Timestamp"])
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K q" + r.get("Queue_Id", "?") + " " + short(r["Kernel_Name"])) for r in seg]
for r in mc:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if t0 <= s <= int(seg[-1]["End_Timestamp"]):
        ev.append((s, e, "COPY " + r.get("Direction", r.get("Name", "?")) + " " + str(r.get("Size", r.get("Bytes", "")))))
ev.sort()
print(f"step of {len(seg)} launches, {(int(seg[-1]['Start_Timestamp']) - t0) / 1e6:.2f} ms; copies in the step: {sum(1 for x in ev if x[2].startswith('COPY'))}")
end = ev[0][1]
last = ev[0]
for s, e, name in ev[1:]:
    if s - end > a.min * 1e3:
        print(f"\n--- idle {(s - end) / 1e3:.1f} us at {(end - t0) / 1e6:.3f} ms   after [{last[2]}]   before [{name}]")
        calls = [r for r in api if int(r["Start_Timestamp"]) < s and int(r["End_Timestamp"]) > end]
        calls.sort(key=lambda r: int(r["Start_Timestamp"]))
        for r in calls[:14]:
            print(f"      host: {r.get('Function', r.get('Name', '?'))[:40]:40s} {(int(r['Start_Timestamp']) - t0) / 1e6:9.3f} -> {(int(r['End_Timestamp']) - t0) / 1e6:9.3f} ms  (tid {r.get('Thread_Id', '?')})")
        if len(calls) > 14:
            print(f"      ... {len(calls)} host calls overlap the gap")
    if e > end:
        end, last = e, (s, e, name)
