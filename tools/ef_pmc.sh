#!/bin/bash
# HBM-traffic and SQ counters per kernel of one EfficientNet-B0 forward+backward (256 crops, side stream off).
# Separate passes per counter group as MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE do not fit one pass);
# counter runs carry no trace flags.  Usage (GPU box): tools/ef_pmc.sh <outdir-under-gpurun_out> [perf script args]
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/$1
script=${2:-perf_ef.py}
mkdir -p $out
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  MT_SIDE_STREAM=0 timeout 900 rocprofv3 --pmc $grp --output-format csv -d $out/$tag -o pmc -- python $GRAFT_REPO_ROOT/tools/$script --bwd --iters 1 2>&1 | grep -E "crops=|B=|rror" | head -3
done
MT_SIDE_STREAM=0 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out/trace -o kt -- python $GRAFT_REPO_ROOT/tools/$script --bwd --iters 1 2>&1 | grep -E "crops=|B="
find $out -name "*.csv" | head -20
