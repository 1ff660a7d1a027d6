#!/bin/bash
# eval-path session: the tests that run eval-mode kernels, then the eval forward's bench line and per-kernel stats
TAG=${1:-evaltrip}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -k "eval or folded or bn_eval or fold or conf or erp or native_seam or regular" 2>&1 | grep -v 'MIOpen\|^add \|^MODE\|^using' | tail -5
bash tools/gpu_eval_prof.sh $TAG
