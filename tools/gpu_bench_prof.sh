#!/bin/bash
# The bench line + the rocprofv3 kernel stats of the same command (no test run): gpurun_out/<tag>/.
TAG=${1:-bp}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1200 python bench.py --steps 5 --warmup 2 > $OUT/bench.log 2>&1 ; grep '^{' $OUT/bench.log > $OUT/bench.json; cut -c1-300 $OUT/bench.json
timeout 600 python bench.py --steps 5 --warmup 2 --conv-arith f32 --no-cpu-baseline --no-eval-b1 2>/dev/null | grep '^{' > $OUT/bench_conv_arith_f32.json; cut -c1-200 $OUT/bench_conv_arith_f32.json
timeout 600 python bench.py --mode eval --batch 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' > $OUT/bench_eval_b1.json; cut -c1-200 $OUT/bench_eval_b1.json
cd /tmp ; timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-eval-b1 --no-collective-self-test > $OUT/rocprof.log 2>&1
cd $R ; for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -1); do python3 tools/profile_summary.py $f 60 > $OUT/profile_summary_per_step.txt; head -14 $OUT/profile_summary_per_step.txt; done
find $OUT/prof -name "*kernel_trace*.csv" -size +20M -delete
