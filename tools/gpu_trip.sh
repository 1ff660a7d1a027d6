#!/bin/bash
# One GPU-box session: the whole -m gpu tier, the bench line, rocprofv3 kernel stats of the same command.  Everything lands in
# gpurun_out/<tag>/.   usage (from the repo root on the GPU box):  bash tools/gpu_trip.sh [tag]
TAG=${1:-r1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
echo "== pytest -m gpu" ; timeout 2400 python -m pytest tests -m gpu -q --timeout 1200 -s 2>&1 | grep -v 'MIOpen\|^add \|^MODE\|^using' > $OUT/pytest_gpu.log ; grep -E '^(FAILED|ERROR)|passed|failed' $OUT/pytest_gpu.log | tail -30
echo "== bench" ; timeout 1200 python bench.py --steps 5 --warmup 2 > $OUT/bench.log 2>&1 ; grep '^{' $OUT/bench.log > $OUT/bench.json; cut -c1-400 $OUT/bench.json
echo "== bench f32" ; timeout 600 python bench.py --steps 5 --warmup 2 --conv-arith f32 --no-cpu-baseline --no-eval-b1 2>/dev/null | grep '^{' > $OUT/bench_conv_arith_f32.json; cut -c1-200 $OUT/bench_conv_arith_f32.json
echo "== rocprofv3 kernel-trace" ; cd /tmp ; timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-eval-b1 > $OUT/rocprof.log 2>&1 ; tail -2 $OUT/rocprof.log | cut -c1-300
cd $R ; for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -1); do head -101 $f > $OUT/kernel_stats_top100.csv; python3 tools/profile_summary.py $f 60 > $OUT/profile_summary_per_step.txt; head -14 $OUT/profile_summary_per_step.txt; done
find $OUT/prof -name "*kernel_trace*.csv" -size +20M -delete
# the inference line (BASELINE configs[1]), the fusion network's line and the driver's smoke call (typed by hand next to this script up to r06zz)
echo "== bench eval / fusion, smoke" ; timeout 600 python bench.py --mode eval --batch 1 --steps 20 --warmup 5 2>/dev/null | grep '^{' > $OUT/bench_eval_b1.json
timeout 600 python bench.py --mode fusion --steps 20 --warmup 3 2>/dev/null | grep '^{' > $OUT/bench_fusion.json
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 > $OUT/smoke.txt ; cat $OUT/smoke.txt
du -sh $OUT
