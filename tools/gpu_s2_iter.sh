#!/bin/bash
# Quick iteration on the stride-2 3-D split kernel: its tests + the bench line's per-kernel timings.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x --timeout 900 -s -k "stride2 or deconv3d or hourglass" 2>&1 | grep -v "MIOpen\|^MODE\|^using\|^add" | tail -${2:-14}
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-eval-b1 2>/dev/null | grep "^{" > gpurun_out/${1:-s2}_bench.json
python - <<PY
import json
d=json.loads(open("gpurun_out/${1:-s2}_bench.json").read())
print(d["value"], d["ms_per_step"])
for k,v in d["kernels"].items():
    if " s2 " in k or "deconv" in k: print(k, v)
PY
