#!/bin/bash
# A/B of an env switch on one box: serialised per-kernel averages of the TimeSformer step (side stream off) with the switch unset / set,
# then interleaved bench lines.  Usage: tools/lab/ab_env.sh ENVNAME regex [nobench|bench] [value when set, default 1]
envn=$1; rx=$2; val=${4:-1}
out=$GRAFT_REPO_ROOT/gpurun_out/ab_$envn; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for v in unset 1; do
  if [ $v = unset ]; then unset $envn; else export $envn=$val; fi
  MT_SIDE_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/p$v -o tsf -- python $GRAFT_REPO_ROOT/tools/perf_tsf.py --bwd --iters 3 2>&1 | grep "B="
  f=$(find $out/p$v -name "*kernel_stats.csv" | head -1)
  echo "== $envn=$v"; python - "$f" "$rx" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rx = re.compile(sys.argv[2])
print("all kernels %.2f ms" % (sum(float(r["TotalDurationNs"]) for r in rows) / 1e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    if rx.search(r["Name"]):
        print("  %8.1f us avg x%4s  %s" % (float(r["AverageNs"]) / 1e3, r["Calls"], r["Name"][:100]))
PY
  rm -f $out/p$v/*kernel_trace.csv
done
cd $GRAFT_REPO_ROOT
[ "$3" = nobench ] && exit 0
for r in 1 2; do for v in unset 1; do
  if [ $v = unset ]; then unset $envn; else export $envn=$val; fi
  echo -n "$envn=$v "; python bench.py --steps 20 --warmup 8 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])"
done; done
