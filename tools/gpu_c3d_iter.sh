#!/bin/bash
# Quick iteration on the stride-1 3-D split kernels: float64 test at small shapes + bench per-kernel timings.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
python -m pytest tests/test_gpu_split.py -m gpu -q -x --timeout 900 -k "split_forward_and_input or split_weight_gradient_is or through_autograd" 2>&1 | tail -2
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-eval-b1 --conv-arith bf16x6 2>/dev/null | grep "^{" > gpurun_out/${1:-c3d}_bench.json
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-eval-b1 --conv-arith f32 --no-kernel-timing 2>/dev/null | grep "^{" > gpurun_out/${1:-c3d}_bench_f32.json
python - <<PY
import json
d=json.loads(open("gpurun_out/${1:-c3d}_bench.json").read()); f=json.loads(open("gpurun_out/${1:-c3d}_bench_f32.json").read())
print("step %.2f ms (f32 mode on this box %.2f ms; ratio %.4f)" % (d["ms_per_step"], f["ms_per_step"], d["ms_per_step"]/f["ms_per_step"]))
for k,v in d["kernels"].items():
    if k.startswith("conv3d") and " s1 " in k and "->1 " not in k: print(k, v)
PY
