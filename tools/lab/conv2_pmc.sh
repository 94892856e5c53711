#!/bin/bash
# SQ counters of the three conv2 GEMMs alone (tools/lab/conv2_lab.py); counter passes carry no trace flags
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/conv2pmc; rm -rf $out; mkdir -p $out
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $grp | cut -d' ' -f1)
  LAB_REPS=1 timeout 300 rocprofv3 --pmc $grp --output-format csv -d $out/$tag -o pmc -- python $GRAFT_REPO_ROOT/tools/lab/conv2_lab.py > $out/$tag.log 2>&1
done
python - <<'PY'
import csv, glob, collections, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/conv2pmc"
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    seen = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        if "gemm_kernel" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        seen[(k, r["Counter_Name"])] += 1
    for (k, c), n in seen.items(): cnt[(k, c)] = n
with open(out + "/summary.txt", "w") as fo:
    for k, d in acc.items():
        fo.write(k + "\n")
        for c, v in sorted(d.items()):
            fo.write(f"   {c:32s} {v / max(cnt[(k, c)], 1):16.0f}  (per launch, {cnt[(k, c)]} launches)\n")
print(open(out + "/summary.txt").read())
PY
