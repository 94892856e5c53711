#!/bin/bash
# config 5: Xception tests, bench lines at a few conv2-wgrad side-stream shares, kernel trace (lab; writes gpurun_out/c5probe/)
out=$GRAFT_REPO_ROOT/gpurun_out/c5probe
rm -rf $out; mkdir -p $out
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_xception.py -x -q 2>&1 | tail -3 > $out/tests.log
for f in 0.7 1.0 0.5; do
  MT_XC_CONV2_WGRAD_SIDE=$f python bench.py --config 5 --steps 5 --warmup 3 --no-cpu-baseline --no-extras 2>$out/c5.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('side $f', d['ms_per_step'], d['value'])" >> $out/lines.txt
done
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/c5_stats -o c5 -- python $GRAFT_REPO_ROOT/bench.py --config 5 --steps 4 --warmup 5 --no-cpu-baseline --no-extras > $out/c5_line_rocprof.json 2>$out/c5_rocprof.err)
python tools/step_timeline.py $(find $out/c5_stats -name "*kernel_trace.csv") > $out/c5_timeline.txt 2>&1
rm -rf $out/c5_stats
cat $out/tests.log $out/lines.txt; head -34 $out/c5_timeline.txt
