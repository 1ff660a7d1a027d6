#!/bin/bash
# GPU session: the repeatability hunt (two processes on the GPU) + the stride-2 / transposed / classifier tests + per-kernel timings of the step.
TAG=${1:-rep}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 1700 python -m pytest tests/test_gpu_repeat.py tests/test_gpu_classif.py tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_split.py -m gpu -q --timeout 1500 -s -k "repeat or second_process or classif or stride2 or deconv or hourglass or transposed or s2" 2>&1 | grep -v 'MIOpen\|^add \|^MODE\|^using' > $OUT/pytest.log
grep -E 'differ|^(FAILED|ERROR)|passed|failed' $OUT/pytest.log | tail -30
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-eval-b1 2>/dev/null | grep "^{" > $OUT/bench.json
python - <<PY
import json
d=json.loads(open("$OUT/bench.json").read())
print(d["value"], d["ms_per_step"], d["targets"])
for k,v in d["kernels"].items():
    if " s2 " in k or "deconv" in k or "classif" in k or "head" in k: print(k, v)
PY
