#!/bin/bash
# Zero-vs-random operand pair of the plane GEMM loop with clocks from counters.  Usage (GPU box): tools/lab/power_pair.sh <outfile-under-gpurun_out>
cd /tmp && export TMPDIR=/tmp
d=/tmp/power_pair; rm -rf $d; mkdir -p $d
out=$GRAFT_REPO_ROOT/gpurun_out/$1
python $GRAFT_REPO_ROOT/tools/lab/power_pair.py > $out 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $d/pmc -o p -- python $GRAFT_REPO_ROOT/tools/lab/power_pair.py > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv -d $d/kt -o k -- python $GRAFT_REPO_ROOT/tools/lab/power_pair.py > /dev/null 2>&1
python - $d >> $out <<'PY'
import csv, glob, sys
from collections import defaultdict
d = sys.argv[1]
pm = glob.glob(d + "/pmc/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob(d + "/kt/**/*kernel_trace.csv", recursive=True)[0]
per = defaultdict(dict); names = {}
for r in csv.DictReader(open(pm)):
    i = int(r["Dispatch_Id"]); per[i][r["Counter_Name"]] = per[i].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"]); names[i] = r["Kernel_Name"]
gemm = [per[i] for i in sorted(per) if "gemm_planes_kernel" in names[i]]
rows = sorted((r for r in csv.DictReader(open(kt)) if "gemm_planes_kernel" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
wall = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
assert len(gemm) == len(wall) == 80, (len(gemm), len(wall))
print("counter pass (rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES) + trace pass (--kernel-trace), launches 11-40 = random, 51-80 = zero operands:")
for tag, sl in (("random", slice(10, 40)), ("zero", slice(50, 80))):
    g = sum(c["GRBM_GUI_ACTIVE"] for c in gemm[sl]) / 30 / 8            # per XCD
    b = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"] for c in gemm[sl]) / 30
    w = sum(wall[sl]) / 30
    print(f"  {tag:6s}: wall {w:7.1f} us (trace pass)  GRBM_GUI_ACTIVE / 8 XCDs {g / 1e3:8.1f} k cycles  MFMA busy cycles {b / 1e6:8.2f} M "
          f"= {b / (g * 1024):.3f} of 1024 SIMDs x active cycles; busy cycles x 1024 FLOP = {b * 1024 / 1e9:.1f} GFLOP (6 x 2 n^3 = {6 * 2 * 4096 ** 3 / 1e9:.1f});"
          f" clock (GUI_ACTIVE / wall) {g / w / 1e3:.2f} GHz")
PY
