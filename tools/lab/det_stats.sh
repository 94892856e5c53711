#!/bin/bash
# Kernel stats of the deterministic step next to the default step (rocprofv3 --kernel-trace --stats).  Usage (GPU box): tools/lab/det_stats.sh <outdir-under-gpurun_out>
out=$GRAFT_REPO_ROOT/gpurun_out/$1
rm -rf $out; mkdir -p $out
for mode in 0 1; do
  (cd /tmp && export TMPDIR=/tmp && MT_DETERMINISTIC=$mode rocprofv3 --kernel-trace --stats --output-format csv -d $out/m$mode -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-extras > $out/line$mode.json 2>$out/err$mode.txt)
  cp $(find $out/m$mode -name "*kernel_stats.csv" | head -1) $out/stats$mode.csv
  rm -rf $out/m$mode
done
python - $out <<'PY'
import csv, re, sys
from collections import defaultdict
out = sys.argv[1]
def load(f):
    d = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        n = re.sub(r"<.*|\(.*", "", r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("mt::", ""))[:48]
        d[n][0] += int(r["Calls"]); d[n][1] += float(r["TotalDurationNs"]) / 1e6
    return d
a, b = load(out + "/stats0.csv"), load(out + "/stats1.csv")
steps = 12.0
rows = sorted(set(a) | set(b), key=lambda k: -(b.get(k, [0, 0])[1] - a.get(k, [0, 0])[1]))
with open(out + "/det_vs_default.txt", "w") as f:
    f.write("kernel-time per step (ms, both queues summed; 12 steps in each trace): default | deterministic | difference\n")
    ta = sum(v[1] for v in a.values()) / steps; tb = sum(v[1] for v in b.values()) / steps
    f.write(f"{'TOTAL':48s} {ta:8.2f} {tb:8.2f} {tb - ta:+8.2f}\n")
    for k in rows:
        x, y = a.get(k, [0, 0.0]), b.get(k, [0, 0.0])
        if abs(y[1] - x[1]) / steps > 0.05:
            f.write(f"{k:48s} {x[1] / steps:8.2f} {y[1] / steps:8.2f} {(y[1] - x[1]) / steps:+8.2f}   calls/step {x[0] / steps:.0f} -> {y[0] / steps:.0f}\n")
PY
