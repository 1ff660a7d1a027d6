# The eval forward with its stride-1 3-D layers on two fp16 pieces against three bf16 pieces (--no-eval-f16), same box, A B A B;
# before that the tests of the new path.
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_f16.py -m gpu -q -x --timeout 900 -s -k "eval_forward_on or eval_kernels_leave" 2>&1 | grep -E "eval forward|passed|failed|Error|assert" | tail -12
python -m pytest tests/test_gpu_split.py tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_parity.py -m gpu -q -x --timeout 900 -k "folded or eval or golden or reference or fixture or captured" 2>&1 | grep -E "passed|failed|^FAILED|^ERROR" | tail -5
for r in 1 2; do for v in "" "--no-eval-f16"; do
  echo "== eval $v (round $r)"
  python bench.py --mode eval --batch 1 --steps 30 --warmup 5 --no-cpu-baseline $v 2>/dev/null | grep "^{" > gpurun_out/eval_ab.json; python - <<'P'
import json
d=json.load(open('gpurun_out/eval_ab.json'))
k=d.get('kernels',{})
c3=sum(v['calls']*v['avg_ms']/2 for n,v in k.items() if n.startswith('conv3d_bn_eval') and ' s1 ' in n)
print('%.3f ms per pair  %.1f pairs/s  stride-1 3-D layers %.3f ms  roofline %s %.3f' % (d['ms_per_step'], d['value'], c3, d.get('roofline',{}).get('kernel'), d.get('roofline',{}).get('frac',0)))
P
done; done
