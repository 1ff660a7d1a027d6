#!/bin/bash
# per-label kernel times of bench.py with and without a flag:  tools/gpu_ab_labels.sh <flag> <label-prefix,label-prefix,...>
FLAG=$1; PFX=$2
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for f in "" "$FLAG"; do
  python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-eval-b1 --no-collective-self-test --profile-steps 3 $f 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']; pf='$PFX'.split(',')
print('== flag [$f]: %.2f ms/step' % d['ms_per_step'])
tot=0
for n,v in k.items():
  if n.startswith(tuple(pf)):
    print('   %-46s calls %3d avg %.4f ms total/step %.3f' % (n, v['calls'], v['avg_ms'], v['calls']*v['avg_ms']/3)); tot+=v['calls']*v['avg_ms']/3
print('   sum of these per step: %.3f ms' % tot)"
done
