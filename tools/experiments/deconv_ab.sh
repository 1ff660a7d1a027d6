#!/bin/bash
# same-box A/B of two builds of the library on the transposed / stride-2 kernels.  The two builds are NOT in the repository: build the
# library from the two source states to compare and copy each mode-2022_amd/mode_hip/libmode_hip.so to
# tools/experiments/libmode_hip_new.so / libmode_hip_dcold.so (*.so is git-ignored) before sending the tree to the GPU box.  The script swaps
# them in turn under the product's name (new, old, new, old) and restores `new` at the end.
for v in new dcold; do
  [ -f tools/experiments/libmode_hip_$v.so ] || { echo "tools/experiments/libmode_hip_$v.so is missing (see the header of this script)"; exit 2; }
done
# the product library is swapped in place below: put it back on ANY exit, not only after a clean run
cp mode-2022_amd/mode_hip/libmode_hip.so /tmp/libmode_hip_product_$$.so
trap 'cp /tmp/libmode_hip_product_$$.so mode-2022_amd/mode_hip/libmode_hip.so; rm -f /tmp/libmode_hip_product_$$.so' EXIT
trap 'exit 130' INT TERM
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for v in new dcold new dcold; do
  cp tools/experiments/libmode_hip_$v.so mode-2022_amd/mode_hip/libmode_hip.so
  python - <<PY
import os, sys, torch
sys.path.insert(0, 'mode-2022_amd'); sys.path.insert(0, '.')
from mode_hip import functional as HF
dev = 'cuda:0'
flush = torch.empty(1 << 27, dtype=torch.float32, device=dev)
def t_ms(fn, n=20):
  for _ in range(3): fn()
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(n): fn()
  b.record(); torch.cuda.synchronize()
  return a.elapsed_time(b) / n
out = []
for (ci, co, D, H, W) in ((32, 64, 48, 256, 128), (64, 64, 24, 128, 64)):
  xl = torch.randn(2, co, D // 2, H // 2, W // 2, device=dev); wt = torch.randn(co, ci, 3, 3, 3, device=dev) * 0.05
  out.append('deconv %d->%d: %.4f' % (co, ci, t_ms(lambda: HF.deconv3d_fwd(xl, wt))))
  gy = torch.randn(2, co, D // 2, H // 2, W // 2, device=dev); w = torch.randn(co, ci, 3, 3, 3, device=dev) * 0.05
  out.append('s2 bwd_data %d->%d: %.4f' % (ci, co, t_ms(lambda: HF.conv3d_bwd_data(gy, w, (2, ci, D, H, W), 2))))
print('$v', ' | '.join(out))
PY
done
