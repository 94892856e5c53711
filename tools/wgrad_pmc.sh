#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the TimeSformer weight-gradient GEMM family (one fwd+bwd of perf_tsf.py, side stream off), for the
# K-range-major XCD mapping on and off.  Usage (GPU box): tools/wgrad_pmc.sh <outdir-under-gpurun_out>
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out
for x in 1 0; do
  for grp in FETCH_SIZE WRITE_SIZE; do
    MT_WGRAD_XCD_K=$x MT_SIDE_STREAM=0 timeout 600 rocprofv3 --pmc $grp --output-format csv -d $out/x$x/$grp -o pmc -- python $GRAFT_REPO_ROOT/tools/perf_tsf.py --bwd --iters 1 > /dev/null 2>&1
  done
  python3 - $out/x$x $x <<'PY'
import csv, glob, sys, re
from collections import defaultdict
d, x = sys.argv[1], sys.argv[2]
tot = {}
for grp in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{d}/{grp}/**/*counter_collection.csv", recursive=True)[0]
    per = defaultdict(float); names = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != grp: continue
        per[int(r["Dispatch_Id"])] += float(r["Counter_Value"]); names[int(r["Dispatch_Id"])] = r["Kernel_Name"]
    # TN split kernels: template args "..., 1, 1, 4," (AL = BL = k-major, EPI_ATOMIC)
    sel = [per[k] for k in sorted(per) if re.search(r"gemm_split_kernel<\d+, \d+, \d+, \d+, 1, 1, 4,", names[k])]
    last = sel[-55:]                      # the last backward pass
    tot[grp] = (sum(last) / len(last) * 1024 * (2.0 if grp == "FETCH_SIZE" else 1.0), len(last))
print(f"XCD_K={x}: wgrad family mean per launch: read {tot['FETCH_SIZE'][0] / 1e6:.1f} MB (FETCH_SIZE x2, gfx950), write {tot['WRITE_SIZE'][0] / 1e6:.1f} MB, "
      f"sum {(tot['FETCH_SIZE'][0] + tot['WRITE_SIZE'][0]) / 1e6:.1f} MB over {tot['FETCH_SIZE'][1]} launches (algorithmic 115.1 MB)")
PY
done
