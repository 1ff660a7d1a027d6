#!/bin/bash
# Short GPU session: a pytest selection (-k expression in $2) only.  usage: bash tools/gpu_quick.sh <tag> "<-k expr>"
TAG=${1:-q}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -k "$2" 2>&1 | grep -v 'MIOpen\|^add \|^MODE\|^using' > $OUT/pytest_gpu.log
grep -E '^(FAILED|ERROR)|passed|failed' $OUT/pytest_gpu.log | tail -40
