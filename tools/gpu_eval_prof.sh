#!/bin/bash
# rocprofv3 kernel stats of the eval forward at one pair (configs[1]): gpurun_out/<tag>/
TAG=${1:-evalprof}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 600 python bench.py --mode eval --batch 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' > $OUT/bench_eval_b1.json; cut -c1-200 $OUT/bench_eval_b1.json
cd /tmp ; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o eval -- python3 $R/bench.py --mode eval --batch 1 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-collective-self-test > $OUT/rocprof.log 2>&1
cd $R ; for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -1); do head -45 $f | cut -d, -f1-4 | cut -c1-150 > $OUT/eval_kernel_stats_top.txt; cat $OUT/eval_kernel_stats_top.txt | head -42; done
find $OUT/prof -name "*kernel_trace*.csv" -size +20M -delete
