#!/bin/bash
# One GPU-box session: parity tests, micro-benchmarks, bench line, rocprofv3 kernel stats.  Everything lands in gpurun_out/.
# usage (from the repo root on the GPU box):  bash tools/gpu_trip.sh [tag]
TAG=${1:-r1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
echo "== pytest -m gpu" ; timeout 1500 python -m pytest tests -m gpu -q --timeout 900 2>&1 | grep -v 'MIOpen\|^add \|^MODE\|^using' > $OUT/pytest_gpu.log ; grep -E '^(FAILED|ERROR|[0-9]+ (passed|failed))|passed|failed' $OUT/pytest_gpu.log | tail -30
echo "== microbench" ; timeout 900 python tools/microbench.py > $OUT/microbench.log 2>&1 ; tail -60 $OUT/microbench.log
echo "== bench" ; timeout 1200 python bench.py --steps 3 --warmup 1 > $OUT/bench.log 2>&1 ; grep -v 'MIOpen' $OUT/bench.log | tail -3 | cut -c1-2500
echo "== rocprofv3 kernel-trace" ; cd /tmp ; timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $OUT/rocprof.log 2>&1 ; tail -3 $OUT/rocprof.log
cd $R ; find $OUT/prof -name "*stats*" | head ; for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -1); do head -40 $f > $OUT/kernel_stats_top40.csv; cat $OUT/kernel_stats_top40.csv | cut -c1-220; done
# keep the merge small: drop the raw trace
find $OUT/prof -name "*kernel_trace*.csv" -size +20M -delete
du -sh $OUT
