#!/bin/bash
# Same-box A/B of a bench.py flag: tools/gpu_ab_flag.sh <flag> [rounds]  -- alternates `bench.py` and `bench.py <flag>`, prints ms per step
FLAG=$1; N=${2:-3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for i in $(seq 1 $N); do
  a=$(python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-eval-b1 --no-kernel-timing --no-collective-self-test 2>/dev/null | grep '^{' | python -c "import json,sys; print('%.2f' % json.loads(sys.stdin.read())['ms_per_step'])")
  b=$(python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-eval-b1 --no-kernel-timing --no-collective-self-test $FLAG 2>/dev/null | grep '^{' | python -c "import json,sys; print('%.2f' % json.loads(sys.stdin.read())['ms_per_step'])")
  echo "default $a ms   with $FLAG $b ms"
done
