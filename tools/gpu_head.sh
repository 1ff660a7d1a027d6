#!/bin/bash
# GPU session: tests touched by the head / loss fusion, then same-box A/B of bench flags.
TAG=${1:-head}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_steps.py tests/test_gpu_classif.py tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q --timeout 900 -s -k "steps or classif or head or forward_loss or eval_pack or captured_eval or native_seam or conv3d_fwd_bwd or graph_replay" 2>&1 | grep -v 'MIOpen\|^add \|^MODE\|^using' > $OUT/pytest.log
grep -E 'losses|running statistics|relative L2|^(FAILED|ERROR)|passed|failed|Error:' $OUT/pytest.log | tail -40
for f in "$@"; do [ "$f" = "$TAG" ] && continue; bash tools/gpu_ab_flag.sh $f 2; done
