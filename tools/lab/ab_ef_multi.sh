#!/bin/bash
# Serialised kernel time of the extractor step (tools/perf_ef.py under rocprofv3, side stream off) for several values of one env switch.
# Usage: tools/lab/ab_ef_multi.sh ENVNAME regex v1 v2 ...
envn=$1; rx=$2; shift 2
out=$GRAFT_REPO_ROOT/gpurun_out/abm_$envn; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  export $envn=$v
  MT_SIDE_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/p$v -o ef -- python $GRAFT_REPO_ROOT/tools/perf_ef.py --bwd --iters 3 2>&1 | grep "crops="
  f=$(find $out/p$v -name "*kernel_stats.csv" | head -1)
  echo "== $envn=$v"; python - "$f" "$rx" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rx = re.compile(sys.argv[2])
n = 5.0
print("all kernels %.2f ms / step;  matching %.2f ms / step" % (sum(float(r["TotalDurationNs"]) for r in rows) / 1e6 / n, sum(float(r["TotalDurationNs"]) for r in rows if rx.search(r["Name"])) / 1e6 / n))
fam = {}
for r in rows:
    if rx.search(r["Name"]):
        k = re.sub(r"<.*", "", r["Name"].replace("void ", "").replace("(anonymous namespace)::", ""))
        fam[k] = fam.get(k, 0.0) + float(r["TotalDurationNs"]) / 1e6 / n
for k, v in sorted(fam.items(), key=lambda kv: -kv[1]):
    print("   %6.2f ms/step  %s" % (v, k))
PY
  rm -f $out/p$v/*kernel_trace.csv
done
