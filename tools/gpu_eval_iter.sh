#!/bin/bash
# Eval path: folded-BatchNorm tests of the 3-D kernels, the eval-mode model tests, and the eval B = 1 bench leg.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_split.py -m gpu -q -x --timeout 900 -k "folded_batchnorm_conv3d or folded_epilogues or split_with_the_folded" 2>&1 | tail -2
[ -n "$QUICK" ] || python -m pytest tests/test_gpu_parity.py tests/test_gpu_model.py -m gpu -q -x --timeout 1200 -k "eval" 2>&1 | tail -2
python bench.py --mode eval --batch 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('eval B=1: %.3f ms per pair' % d['ms_per_step']); [print('  ', k, v['avg_ms'], v['TFLOPs']) for k, v in d.get('kernels', {}).items() if 's2 ' in k or 'deconv' in k]"
