#!/bin/bash
# BatchNorm traversal orders (MODE_BN_ORDER bit mask, csrc/bn_act.hip) inside the training step: same box, one run each
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for O in 0 1 5 15 4 2 0; do
  MODE_BN_ORDER=$O python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-eval-b1 --no-collective-self-test 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['kernels']
f=sum(v['calls']*v['avg_ms'] for n,v in k.items() if n.startswith('bn_train_fwd'))/2
b=sum(v['calls']*v['avg_ms'] for n,v in k.items() if n.startswith('bn_train_bwd'))/2
print('order %2s: %.2f ms/step   bn_train_fwd %.2f ms/step (big 3-D layer %.4f)   bn_train_bwd %.2f (%.4f)' % (sys.argv[1], d['ms_per_step'], f, k['bn_train_fwd[2x32 48x256x128]']['avg_ms'], b, k['bn_train_bwd[2x32 48x256x128]']['avg_ms']))
" $O
done
