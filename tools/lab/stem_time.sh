#!/bin/bash
# stem forward: streaming MFMA kernel vs the im2col GEMM (kernel trace of an EfficientNet forward, 256 crops)
cd /tmp && export TMPDIR=/tmp
for s in 1 0; do
  out=$GRAFT_REPO_ROOT/gpurun_out/stem/s$s; rm -rf $out; mkdir -p $out
  MT_STEM_DIRECT=$s timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python $GRAFT_REPO_ROOT/tools/perf_ef.py --iters 3 > /dev/null 2>&1
  python - $out <<'PY'
import csv,sys,glob
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name']
    if 'stem_mfma' in n or ('gemm_kernel<4, 1, 2, 1, 0, 0, 5' in n): print(sys.argv[1][-2:], n[:60], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
PY
done
