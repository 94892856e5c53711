#!/bin/bash
# effective shader clock (GRBM_GUI_ACTIVE / 8 XCDs / duration) of the lab GEMMs, full vs ablated builds
cd /tmp && export TMPDIR=/tmp
for ab in 0 9 15; do
  out=$GRAFT_REPO_ROOT/gpurun_out/clk_$ab
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $out -o p -- $GRAFT_REPO_ROOT/tools/lab/gemm_lab_ab$ab 12 "$1" > /dev/null 2>&1
  python3 - <<PY
import csv,glob,collections
f=glob.glob("$out/**/*counter_collection.csv", recursive=True)[0]
per=collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    d=r["Dispatch_Id"]; per[d][r["Counter_Name"]]=float(r["Counter_Value"]); per[d]["t"]=int(r["End_Timestamp"])-int(r["Start_Timestamp"]); per[d]["n"]=r["Kernel_Name"][:60]
agg=collections.defaultdict(list)
for d,v in per.items():
    if v["t"]>50000 and "GRBM_GUI_ACTIVE" in v: agg[v["n"]].append((v["GRBM_GUI_ACTIVE"]/8/v["t"], v["t"]/1e3))
print("ablate $ab")
for n,l in agg.items():
    l=l[len(l)//2:]
    print("   %-62s clock %.3f GHz  %.1f us" % (n, sum(x for x,_ in l)/len(l), sum(y for _,y in l)/len(l)))
PY
done
