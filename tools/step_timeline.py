#!/usr/bin/env python3
"""In-step occupancy of the GPU from a rocprofv3 --kernel-trace of bench.py: for the last full step, the union of kernel
intervals (device busy), the idle gaps, per-queue busy time, overlap (>= 2 kernels in flight) and the families' in-step time."""
import argparse, csv, re
from collections import defaultdict

ap = argparse.ArgumentParser()
ap.add_argument("csv")
ap.add_argument("--start", default="stem_mfma_kernel", help="substring of the kernel that starts a step (the stem kernel)")
ap.add_argument("--steps-from-end", type=int, default=0,
                help="which step of the trace: bench.py ends with three enqueue-timing steps that follow a synchronize (the host is not "
                     "ahead there and its launch latency shows as device idle); 4-9 from the end lie inside the timed region of "
                     "`bench.py --steps 8`.  0 (default): the one of those six with the median span (a single step can catch a host "
                     "hiccup of the profiler: one evidence run showed a 2.9 ms gap in step 6 and none in its neighbours)")
ap.add_argument("--top", type=int, default=40, help="symbols listed per queue")
a = ap.parse_args()
rows = sorted(csv.DictReader(open(a.csv)), key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "").replace("mt::", "")
    return re.sub(r"\(.*", "", n)[:60]


starts = [i for i, r in enumerate(rows) if a.start in r["Kernel_Name"]]
if a.steps_from_end == 0:
    cand = [k for k in range(4, 10) if k < len(starts)]
    spans = sorted((int(rows[starts[-k + 1]]["Start_Timestamp"]) - int(rows[starts[-k]]["Start_Timestamp"]), k) for k in cand)
    a.steps_from_end = spans[len(spans) // 2][1]
    print(f"(step {a.steps_from_end} from the end: median span of steps 4-9 from the end, start-to-start spans "
          f"{[round(sp / 1e6, 2) for sp, _ in sorted(spans, key=lambda x: x[1])]} ms)")
lo, hi = starts[-a.steps_from_end], starts[-a.steps_from_end + 1]
seg = rows[lo:hi]
t0 = int(seg[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in seg)
ev = []
for r in seg:
    ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
busy = over = 0; depth = 0; last = t0; gaps = []
for t, d in ev:
    if depth >= 1: busy += t - last
    if depth >= 2: over += t - last
    if depth == 0 and t - last > 3000: gaps.append((t - last, last - t0))
    depth += d; last = t
qs = defaultdict(float)
for r in seg:
    qs[r.get("Queue_Id", "?")] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print(f"step span {(t1 - t0) / 1e6:.2f} ms; device busy {busy / 1e6:.2f} ms ({100 * busy / (t1 - t0):.1f} %); >=2 kernels in flight {over / 1e6:.2f} ms; {len(seg)} launches")
print("per-queue kernel time (ms):", {k: round(v / 1e6, 2) for k, v in qs.items()})
print(f"idle gaps > 3 us: {len(gaps)} totalling {sum(g for g, _ in gaps) / 1e6:.2f} ms; largest:", [(round(g / 1e3, 1), round(at / 1e6, 2)) for g, at in sorted(gaps, reverse=True)[:8]])
fam = defaultdict(lambda: [0, 0.0])
for r in seg:
    f = fam[short(r["Kernel_Name"])]
    f[0] += 1; f[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("in-step kernel time by symbol (us):")
for k, (n, d) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"  {d / 1e3:7.2f} ms {n:4d}  {k}")
for q in sorted(qs, key=lambda k: -qs[k]):
    qf = defaultdict(lambda: [0, 0.0])
    for r in seg:
        if r.get("Queue_Id", "?") != q:
            continue
        f = qf[short(r["Kernel_Name"])]
        f[0] += 1; f[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f"queue {q}: {qs[q] / 1e6:.2f} ms of kernel time; by symbol:")
    for k, (n, d) in sorted(qf.items(), key=lambda kv: -kv[1][1])[:a.top]:
        print(f"  {d / 1e3:7.2f} ms {n:4d}  {k}")
