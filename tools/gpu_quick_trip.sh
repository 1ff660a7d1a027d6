#!/bin/bash
# quick session: a selection of the GPU tests (first argument: pytest -k expression), the bench line and the per-kernel stats of the step
TAG=${1:-quick}
SEL=${2:-"sphere or conv3d or split"}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -k "$SEL" 2>&1 | grep -v 'MIOpen\|^add \|^MODE\|^using' | tail -8
timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench.log 2>&1 ; grep '^{' $OUT/bench.log > $OUT/bench.json; cut -c1-200 $OUT/bench.json
cd /tmp ; timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-eval-b1 --no-collective-self-test > $OUT/rocprof.log 2>&1
cd $R ; for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -1); do python3 tools/profile_summary.py $f 60 > $OUT/profile_summary_per_step.txt; head -24 $OUT/profile_summary_per_step.txt; done
find $OUT/prof -name "*kernel_trace*.csv" -size +20M -delete
