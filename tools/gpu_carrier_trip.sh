#!/bin/bash
# the gradient-carrier tests, then the bench line with and without them on the same box
TAG=${1:-carrier}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -k "carrier or gradient_already or folded or steps or train_forward_backward or graph_replay or sinks" 2>&1 | grep -v 'MIOpen\|^add \|^MODE\|^using' | tail -12
for i in 1 2; do
  a=$(timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-eval-b1 --no-collective-self-test 2>/dev/null | grep '^{' | python3 -c "import sys,json; print('%.2f' % json.loads(sys.stdin.read())['ms_per_step'])")
  b=$(timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-eval-b1 --no-collective-self-test --no-grad-carriers 2>/dev/null | grep '^{' | python3 -c "import sys,json; print('%.2f' % json.loads(sys.stdin.read())['ms_per_step'])")
  echo "default $a ms   with --no-grad-carriers $b ms"
done
