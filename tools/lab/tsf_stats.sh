#!/bin/bash
# Serialised per-kernel averages of the TimeSformer step (side stream off) + two bench lines.  Usage: tools/lab/tsf_stats.sh tag regex
tag=$1; rx=$2
out=$GRAFT_REPO_ROOT/gpurun_out/tsf_$tag; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
MT_SIDE_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o tsf -- python $GRAFT_REPO_ROOT/tools/perf_tsf.py --bwd --iters 3 2>&1 | grep "B="
f=$(find $out -name "*kernel_stats.csv" | head -1)
python - "$f" "$rx" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rx = re.compile(sys.argv[2])
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("all kernels %.2f ms" % (tot / 1e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    if rx.search(r["Name"]):
        print("  %8.1f us avg x%4s  %s" % (float(r["AverageNs"]) / 1e3, r["Calls"], r["Name"][:100]))
PY
rm -f $out/*kernel_trace.csv
cd $GRAFT_REPO_ROOT
for r in 1 2; do python bench.py --steps 20 --warmup 8 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])"; done
