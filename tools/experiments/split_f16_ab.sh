#!/bin/bash
# same-box A/B of the stride-1 3-D split kernel in its product arithmetic (three bf16 pieces, six MFMAs per product) against the
# experiment -DMODE_SPLIT_F16=1 (two fp16 pieces, three MFMAs per product, no scale): time per launch at the
# benchmark volumes, and the error of both against float64 on a small volume.  Build both libraries first
# (MODE_HIP_DEFINES="MODE_SPLIT_F16=1" python mode-2022_amd/mode_hip/build.py -> tools/experiments/libmode_hip_f16x3.so; the default
# build -> ..._bf16x6.so; *.so is git-ignored).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for v in bf16x6 f16x3; do
  [ -f tools/experiments/libmode_hip_$v.so ] || { echo "tools/experiments/libmode_hip_$v.so is missing (see the header of this script)"; exit 2; }
done
# the product library is swapped in place below: put it back on ANY exit, not only after a clean run
cp mode-2022_amd/mode_hip/libmode_hip.so /tmp/libmode_hip_product_$$.so
trap 'cp /tmp/libmode_hip_product_$$.so mode-2022_amd/mode_hip/libmode_hip.so; rm -f /tmp/libmode_hip_product_$$.so' EXIT
trap 'exit 130' INT TERM
for v in bf16x6 f16x3 bf16x6 f16x3; do
  cp tools/experiments/libmode_hip_$v.so mode-2022_amd/mode_hip/libmode_hip.so
  python - <<PY
import sys, torch
sys.path.insert(0, 'mode-2022_amd'); sys.path.insert(0, '.')
from mode_hip import functional as HF
import torch.nn.functional as F
dev = 'cuda:0'
def t_ms(fn, n=20):
  for _ in range(3): fn()
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(n): fn()
  b.record(); torch.cuda.synchronize()
  return a.elapsed_time(b) / n
out = []
torch.manual_seed(0)
for (c, D, H, W) in ((32, 48, 256, 128), (64, 24, 128, 64)):
  x = torch.randn(2, c, D, H, W, device=dev); w = torch.randn(c, c, 3, 3, 3, device=dev) * 0.05
  out.append('%d@%dx%dx%d fwd %.4f bwd_data %.4f' % (c, D, H, W, t_ms(lambda: HF.conv3d_fwd(x, w, 1)), t_ms(lambda: HF.conv3d_bwd_data(x, w, x.shape, 1))))
# accuracy against float64: random data, and data with a wide dynamic range (activations after a ReLU times a smooth envelope over 6 decades)
for name, scale in (('randn', None), ('6 decades', 6.0), ('gradient-sized (x 1e-7)', -1.0)):
  x = torch.randn(1, 32, 6, 20, 40, device=dev); w = torch.randn(32, 32, 3, 3, 3, device=dev) * 0.05
  if scale and scale > 0:
    env = torch.logspace(0, -scale, 40, device=dev).view(1, 1, 1, 1, 40)
    x = torch.relu(x) * env
  elif scale:
    x = x * 1e-7
  ref = F.conv3d(x.double().cpu(), w.double().cpu(), None, 1, 1)
  got = HF.conv3d_fwd(x, w, 1).double().cpu()
  f32 = F.conv3d(x.cpu(), w.cpu(), None, 1, 1).double()
  err, e32 = (got - ref).abs(), (f32 - ref).abs()
  rel = (err / ref.abs().clamp_min(1e-30))
  out.append('%s: max err %.2e (torch fp32 conv %.2e) of max |y| %.2e; worst relative error where |y| > 1e-6 max: %.2e (fp32: %.2e)' % (
      name, float(err.max()), float(e32.max()), float(ref.abs().max()),
      float(rel[ref.abs() > 1e-6 * ref.abs().max()].max()), float((e32 / ref.abs().clamp_min(1e-30))[ref.abs() > 1e-6 * ref.abs().max()].max())))
print('$v', ' | '.join(out))
PY
done
