#!/bin/bash
# SQ counters of the kernels matching a regex in one TimeSformer step.  Usage: tools/lab/pmc_kernel.sh regex [ENV=VAL ...]
rx=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_kernel; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
env "$@" MT_SIDE_STREAM=0 timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $out -o pmc -- python $GRAFT_REPO_ROOT/tools/perf_tsf.py --bwd --iters 1 2>&1 | grep -E "B=|rror" | head -3
python - "$(find $out -name '*counter_collection.csv' | head -1)" "$rx" <<'PY'
import csv, re, sys, collections
rx = re.compile(sys.argv[2])
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    if not rx.search(r["Kernel_Name"]): continue
    k = r["Kernel_Name"][:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (k, r["Dispatch_Id"])
    if key not in seen: seen.add(key); cnt[k] += 1
for k, d in acc.items():
    print(k, "x", cnt[k], "LDS", r.get("LDS_Block_Size"), "wg", r.get("Workgroup_Size"))
    for c, v in sorted(d.items()): print("   %-28s %14.0f per launch" % (c, v / cnt[k]))
PY
