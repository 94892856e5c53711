#!/bin/bash
# Round-6 evidence run on the GPU box: kernel stats of the default bench.py, per-kernel PMC of the EfficientNet and TimeSformer
# steps (separate FETCH_SIZE / WRITE_SIZE / SQ passes), the family JSON bench.py reads, and the config-2 / config-5 / ragged lines.
# Usage: tools/final_profiles.sh   (writes under gpurun_out/final/, copy what is to be judged into profiles/)
out=$GRAFT_REPO_ROOT/gpurun_out/final
rm -rf $out/bench_stats; mkdir -p $out
cd $GRAFT_REPO_ROOT   # (8 + 4 steps under the profiler: with fewer the host is not yet ahead of the device in the analysed step)
bash tools/ef_pmc.sh final/ef perf_ef.py > $out/ef_pmc.log 2>&1
bash tools/ef_pmc.sh final/tsf perf_tsf.py > $out/tsf_pmc.log 2>&1
python tools/pmc_report.py gpurun_out/final/ef --min 100 --json-out $out/r06_pmc_families.json --kind ef > $out/r06_effnet_pmc_per_kernel.txt 2>&1
python tools/pmc_report.py gpurun_out/final/tsf --min 100 --start embed_fwd_kernel --json-out $out/r06_pmc_families.json --kind tsf > $out/r06_tsf_pmc_per_kernel.txt 2>&1
cp $out/r06_pmc_families.json profiles/r06_pmc_families.json 2>/dev/null
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench_stats -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-extras > $out/r06_bench_b32_line_under_rocprof.json 2>$out/bench_rocprof.err)
python tools/step_timeline.py $(find $out/bench_stats -name "*kernel_trace.csv") > $out/r06_bench_b32_in_step_timeline.txt 2>&1
cp $(find $out/bench_stats -name "*kernel_stats.csv" | head -1) $out/r06_bench_b32_kernel_stats.csv 2>/dev/null
python bench.py --steps 20 --warmup 5 > $out/r06_bench_b32_line.json 2>$out/bench.err
python bench.py --config 2 --steps 10 --warmup 5 --no-cpu-baseline > $out/r06_bench_config2_b16_line.json 2>/dev/null
python bench.py --config 5 --steps 5 --warmup 3 --no-cpu-baseline > $out/r06_bench_config5_xs_line.json 2>/dev/null
python bench.py --ragged --steps 10 --warmup 5 --no-cpu-baseline --no-extras > $out/r06_bench_b32_ragged_line.json 2>/dev/null
rm -rf $out/c5_stats
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/c5_stats -o c5 -- python $GRAFT_REPO_ROOT/bench.py --config 5 --steps 4 --warmup 5 --no-cpu-baseline --no-extras > $out/r06_bench_config5_xs_line_under_rocprof.json 2>$out/c5_rocprof.err)
python tools/step_timeline.py $(find $out/c5_stats -name "*kernel_trace.csv") > $out/r06_config5_in_step_timeline.txt 2>&1
cp $(find $out/c5_stats -name "*kernel_stats.csv" | head -1) $out/r06_config5_kernel_stats.csv 2>/dev/null
rm -rf $out/c5_stats $out/bench_stats
ls -la $out | head -40
tail -c 600 $out/r06_bench_b32_line.json
