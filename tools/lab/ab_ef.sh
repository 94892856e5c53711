#!/bin/bash
# A/B of an env switch on one box, EfficientNet side: serialised per-kernel averages of the extractor's forward + backward (side stream
# off, 256 crops) with the switch at value A and value B, then interleaved bench lines.
# Usage: tools/lab/ab_ef.sh ENVNAME A B regex [nobench|bench]
envn=$1; va=$2; vb=$3; rx=$4
out=$GRAFT_REPO_ROOT/gpurun_out/abef_$envn; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for v in $va $vb; do
  export $envn=$v
  MT_SIDE_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/p$v -o ef -- python $GRAFT_REPO_ROOT/tools/perf_ef.py --bwd --iters 3 2>&1 | grep "crops="
  f=$(find $out/p$v -name "*kernel_stats.csv" | head -1)
  echo "== $envn=$v"; python - "$f" "$rx" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rx = re.compile(sys.argv[2])
n = 5.0   # perf_ef.py: 2 warm-up + 3 timed steps
print("all kernels %.2f ms / step" % (sum(float(r["TotalDurationNs"]) for r in rows) / 1e6 / n))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    if rx.search(r["Name"]):
        print("  %8.1f us avg x%4s  %8.2f ms/step  %s" % (float(r["AverageNs"]) / 1e3, r["Calls"], float(r["TotalDurationNs"]) / 1e6 / n, re.sub(r"\(.*", "", r["Name"].replace("void ", "").replace("(anonymous namespace)::", ""))[:90]))
PY
  rm -f $out/p$v/*kernel_trace.csv
done
cd $GRAFT_REPO_ROOT
[ "$5" = nobench ] && exit 0
for r in 1 2; do for v in $va $vb; do
  export $envn=$v
  echo -n "$envn=$v "; python bench.py --steps 20 --warmup 8 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])"
done; done
