#!/bin/bash
# Quick iteration on the spherical split kernels: float64 tests at the benchmark shape + kernel durations.
TAG=${1:-it}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_split.py -m gpu -q --timeout 600 -s -k "split_sphere and (128-256-2 or 64-128 or 32-64 or 128-256-1)" 2>&1 | grep -v 'MIOpen\|^add \|^MODE\|^using' > $OUT/pytest.log
grep -E "sphere_conv_bwd|FAILED|ERROR|passed|failed|Error" $OUT/pytest.log | cut -c1-260
cd /tmp
for C in sphere_bwd_weight_t sphere_bwd_data_t; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${C}_trace -o t -- python3 $R/tools/one_kernel.py $C --no-flush > $OUT/${C}_trace.log 2>&1
  python3 - <<PY
import csv, glob, os
for f in glob.glob(os.path.join('$OUT', '${C}_trace', '**', '*kernel_stats.csv'), recursive=True):
  for row in list(csv.DictReader(open(f)))[:4]:
    print('  $C  %-60s calls %4s avg %9.1f us' % (row['Name'][:60], row['Calls'], float(row['AverageNs']) / 1e3))
PY
done
find $OUT -name "*kernel_trace.csv" -delete
