#!/bin/bash
# One bench run boiled down to a few lines: the step time and the per-step time of the kernel families (for tools/ab_libs.sh).
#   usage: bash tools/bench_brief.sh [label-regex] [bench.py arguments ...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
PAT=${1:-'^$'}; shift
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-eval-b1 --no-collective-self-test "$@" 2>/dev/null | grep "^{" > /tmp/bench_brief_$$.json
python - "$PAT" /tmp/bench_brief_$$.json <<'PY'
import json, re, sys
pat, path = sys.argv[1], sys.argv[2]
d = json.loads(open(path).read())
k = d.get('kernels', {})
steps = 2.0
fam = {}
for name, v in k.items():
  f = ('conv3d_s1' if re.match(r'conv3d_\w+\[\d+->(32|64) s1', name) else 'conv3d_s2+deconv' if re.match(r'(conv3d_\w+\[.* s2|deconv3d)', name) else
       'sphere' if name.startswith('sphere') else 'bn' if name.startswith('bn_') else 'conv2d' if name.startswith(('conv2d', 'conv1x1', 'conv_stem')) else
       'classif' if name.startswith(('classif', 'conv3d')) else 'head' if name.startswith(('head', 'smooth')) else 'other')
  fam[f] = fam.get(f, 0.0) + v['calls'] * v['avg_ms'] / steps
print('step %.2f ms | ' % d['ms_per_step'] + '  '.join('%s %.2f' % (f, t) for f, t in sorted(fam.items(), key=lambda x: -x[1])))
for name, v in k.items():
  if re.search(pat, name):
    print('   %-52s calls %3d  %.4f ms  %.1f TF  %.0f GB/s' % (name, v['calls'], v['avg_ms'], v['TFLOPs'], v['GBps']))
PY
rm -f /tmp/bench_brief_$$.json
