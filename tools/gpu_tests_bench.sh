#!/bin/bash
# GPU session: the whole -m gpu tier + one bench line.  usage: bash tools/gpu_tests_bench.sh <tag> [pytest -k expr]
TAG=${1:-t}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
if [ -n "$2" ]; then K=(-k "$2"); else K=(); fi
timeout 2400 python -m pytest tests -m gpu -q --timeout 1200 -s "${K[@]}" 2>&1 | grep -v 'MIOpen\|^add \|^MODE\|^using' > $OUT/pytest_gpu.log
grep -E '^(FAILED|ERROR)|passed|failed|peaked|two ranks' $OUT/pytest_gpu.log | tail -60
echo "== bench" ; timeout 1200 python bench.py --steps 5 --warmup 2 > $OUT/bench.log 2>&1 ; grep -v 'MIOpen' $OUT/bench.log | tail -2 | cut -c1-6000
