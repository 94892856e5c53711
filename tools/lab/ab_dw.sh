#!/bin/bash
# A/B of an env switch on one box: per-kernel totals of the EfficientNet step under rocprofv3 + interleaved bench lines.
# Usage: tools/lab/ab_dw.sh ENVNAME kernel_regex
envn=$1; rx=$2
out=$GRAFT_REPO_ROOT/gpurun_out/ab_$envn; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  env $envn=$v MT_SIDE_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/p$v -o ef -- python $GRAFT_REPO_ROOT/tools/perf_ef.py --bwd --iters 3 2>&1 | grep crops=
  f=$(find $out/p$v -name "*kernel_stats.csv" | head -1)
  echo "== $envn=$v"; python - "$f" "$rx" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rx = re.compile(sys.argv[2])
tot = sum(float(r["TotalDurationNs"]) for r in rows)
sel = [r for r in rows if rx.search(r["Name"])]
s = sum(float(r["TotalDurationNs"]) for r in sel)
print("all kernels %.2f ms, matching %.2f ms over %d calls" % (tot / 1e6, s / 1e6, sum(int(r["Calls"]) for r in sel)))
for r in sorted(sel, key=lambda r: -float(r["TotalDurationNs"]))[:12]:
    print("  %8.1f us avg x%4s  %s" % (float(r["AverageNs"]) / 1e3, r["Calls"], r["Name"][:110]))
PY
done
cd $GRAFT_REPO_ROOT
for r in 1 2; do for v in 0 1; do
  echo -n "$envn=$v "; env $envn=$v python bench.py --steps 20 --warmup 8 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])"
done; done
