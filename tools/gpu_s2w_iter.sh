#!/bin/bash
# Iteration on the stride-2 weight-gradient split kernel: its float64 tests, then isolated timings beside the fp32 kernel.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python -m pytest tests/test_gpu_split.py -m gpu -q -x --timeout 900 -k "stride2_weight or transposed_convolution_weight or through_autograd" -s 2>&1 | grep -v "^$" | tail -14
python tools/time_c3d.py s2w 2>&1 | tail -1 | tr '|' '\n'
