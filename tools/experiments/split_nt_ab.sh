#!/bin/bash
# same-box A/B of the stride-1 3-D split kernel with one wave per SIMD (256 threads, four rows per wave) against two waves per SIMD (512
# threads, two rows each: -DMODE_SPLIT_NT=512).  Build both libraries first (MODE_HIP_DEFINES="MODE_SPLIT_NT=512" python
# mode-2022_amd/mode_hip/build.py; copy the .so to tools/experiments/libmode_hip_nt512.so, the default build to ..._nt256.so; *.so is
# git-ignored).  Prints ms per launch of forward / input gradient at the benchmark volume and a small one, and checks one against the other.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for v in nt256 nt512; do
  [ -f tools/experiments/libmode_hip_$v.so ] || { echo "tools/experiments/libmode_hip_$v.so is missing (see the header of this script)"; exit 2; }
done
# the product library is swapped in place below: put it back on ANY exit, not only after a clean run
cp mode-2022_amd/mode_hip/libmode_hip.so /tmp/libmode_hip_product_$$.so
trap 'cp /tmp/libmode_hip_product_$$.so mode-2022_amd/mode_hip/libmode_hip.so; rm -f /tmp/libmode_hip_product_$$.so' EXIT
trap 'exit 130' INT TERM
for v in nt256 nt512 nt256 nt512; do
  cp tools/experiments/libmode_hip_$v.so mode-2022_amd/mode_hip/libmode_hip.so
  python - <<PY
import sys, torch
sys.path.insert(0, 'mode-2022_amd'); sys.path.insert(0, '.')
from mode_hip import functional as HF
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
for (c, D, H, W) in ((32, 48, 256, 128), (64, 24, 128, 64), (64, 12, 64, 32)):
  x = torch.randn(2, c, D, H, W, device=dev); w = torch.randn(c, c, 3, 3, 3, device=dev) * 0.05
  out.append('%d@%dx%dx%d fwd %.4f bwd_data %.4f' % (c, D, H, W, t_ms(lambda: HF.conv3d_fwd(x, w, 1)), t_ms(lambda: HF.conv3d_bwd_data(x, w, x.shape, 1))))
  if c == 64 and D == 12:
    torch.save(HF.conv3d_fwd(x, w, 1).cpu(), '/tmp/split_nt_$v.pt')
print('$v', ' | '.join(out))
PY
done
python - <<PY
import torch
a, b = torch.load('/tmp/split_nt_nt256.pt'), torch.load('/tmp/split_nt_nt512.pt')
print('max |nt256 - nt512| =', float((a - b).abs().max()), 'of', float(a.abs().max()))
PY
