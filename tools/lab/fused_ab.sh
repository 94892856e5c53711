#!/bin/bash
# fused expand-conv backward: z streamed vs folded (MT_EXPAND_REBUILD); serialised launches (MT_SIDE_STREAM=0).
cd /tmp && export TMPDIR=/tmp
for rb in 0 1; do
  out=$GRAFT_REPO_ROOT/gpurun_out/rb/p$rb; mkdir -p $out
  MT_SIDE_STREAM=0 MT_EXPAND_REBUILD=$rb timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python $GRAFT_REPO_ROOT/tools/perf_ef.py --bwd --iters 3 > /dev/null 2>&1
  echo "fold=$rb"; python - $out <<'PY'
import csv,sys,glob,collections
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'bwd_fused' in r['Kernel_Name']: d[r['Kernel_Name'][40:80]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in d.items(): print('  ',k,len(v),[round(x) for x in v[-4:]])
PY
done
