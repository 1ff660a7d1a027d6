#!/bin/bash
# GPU session for the fused classifier head: its tests, the head's micro-benchmark with per-kernel durations, then a same-box A/B of the step.
TAG=${1:-classif}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_classif.py -q --timeout 900 -s -x 2>&1 | grep -v 'MIOpen\|^add \|^MODE\|^using' > $OUT/pytest.log
grep -E 'max err|relative L2|left out|^(FAILED|ERROR)|passed|failed|Error|error' $OUT/pytest.log | tail -60
python tools/experiments/classif_bench.py 2>&1 | tail -3
cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o cb -- python3 $R/tools/experiments/classif_bench.py > $OUT/rocprof.log 2>&1
cd $R; f=$(find $OUT/prof -name "*kernel_stats*.csv" | head -1); python3 - "$f" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
  print('%-70s calls %4s avg %9.1f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3))
P
find $OUT/prof -name "*kernel_trace*.csv" -size +20M -delete
[ -n "$2" ] && bash tools/gpu_ab_flag.sh --no-fused-classif 2
