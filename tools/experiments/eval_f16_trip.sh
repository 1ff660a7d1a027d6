# The eval forward with its stride-1 3-D and 3 x 3 layers on two fp16 pieces against three bf16 pieces (--no-eval-f16), same box, A B A B;
# before that the tests of the path.
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_f16.py -m gpu -q -x --timeout 900 -s -k "eval" 2>&1 | grep -E "eval forward|3x3 .*unit|passed|failed|Error|assert" | tail -16
python -m pytest tests/test_gpu_split.py tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_parity.py tests/test_fusion.py -m gpu -q -x --timeout 900 -k "folded or eval or golden or reference or fixture or captured or fusion or conv2d" 2>&1 | grep -E "passed|failed|^FAILED|^ERROR|Error" | tail -5
for r in 1 2; do for v in "" "--no-eval-f16"; do
  echo "== eval $v (round $r)"
  python bench.py --mode eval --batch 1 --steps 30 --warmup 5 --no-cpu-baseline $v 2>/dev/null | grep "^{" > gpurun_out/eval_ab.json; python - <<'P'
import json
d=json.load(open('gpurun_out/eval_ab.json'))
k=d.get('kernels',{})
c3=sum(v['calls']*v['avg_ms']/2 for n,v in k.items() if n.startswith('conv3d_bn_eval') and ' s1 ' in n)
c2=sum(v['calls']*v['avg_ms']/2 for n,v in k.items() if n.startswith('conv2d_bn_eval'))
print('%.3f ms per pair  %.1f pairs/s  stride-1 3-D layers %.3f ms  3x3 layers %.3f ms  roofline %s %.3f' % (d['ms_per_step'], d['value'], c3, c2, d.get('roofline',{}).get('kernel'), d.get('roofline',{}).get('frac',0)))
P
done; done
python bench.py --mode fusion --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | grep "^{" | cut -c1-200
