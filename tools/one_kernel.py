#!/usr/bin/env python3
"""Runs ONE kernel shape a few times, for PMC passes that must not mix shapes:

    rocprofv3 --pmc FETCH_SIZE -- python3 tools/one_kernel.py <case> [--no-flush]

Between the launches a 1 GiB fill evicts the 256 MiB Infinity Cache (MI355X_MICROARCH.md: launches that re-read the tensors of the
previous launch otherwise find part of them on-die, and FETCH_SIZE comes out below the compulsory bytes).  Cases (batch = what the
benchmark step launches): conv3d_{fwd,bwd_data,bwd_weight}_32 (32->32 at 48x256x128, B=2; bwd_weight = the bench's roofline
kernel), sphere_{fwd,bwd_data,bwd_weight}_t (128->128 at 256x128, 4 images, plane-transposed storage), cost_volume_fwd (B=2),
bn3d_32 (BatchNorm3d(32)+ReLU train fwd+bwd on the 48x256x128 volume, B=2), conv3d_fwd_s2 / conv3d_bwd_data_s2 / conv3d_bwd_weight_s2 (32->64 stride 2 at
48x256x128, B=2), deconv3d_fwd_64_32 (ConvTranspose3d 64->32 at 24x128x64, B=2)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'mode-2022_amd')):
  sys.path.insert(0, p)
import torch  # noqa: E402

from mode_hip import functional as HF  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else 'conv3d_bwd_weight_32'
flush = '--no-flush' not in sys.argv
dev = torch.device('cuda', 0)
thrash = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=dev) if flush else None


def run(fn, n=4):
  for _ in range(n):
    if flush:
      thrash.fill_(1.0)
    fn()
  torch.cuda.synchronize()


if what in ('conv3d_fwd_s2', 'conv3d_bwd_data_s2', 'conv3d_bwd_weight_s2', 'conv3d_bwd_weight_s2_64', 'deconv3d_fwd_64_32'):
  # hourglass conv1 (32 -> 64, stride 2) forward / input gradient at the 48 x 256 x 128 volume, conv6 (ConvTranspose3d 64 -> 32), B = 2
  if what == 'deconv3d_fwd_64_32':
    x = torch.randn(2, 64, 24, 128, 64, device=dev)
    w = torch.randn(64, 32, 3, 3, 3, device=dev) * 0.05
    run(lambda: HF.deconv3d_fwd(x, w))
  elif what == 'conv3d_bwd_weight_s2_64':  # hourglass conv3 (64 -> 64, stride 2) at the 24 x 128 x 64 volume
    x = torch.randn(2, 64, 24, 128, 64, device=dev)
    gy = torch.randn(2, 64, 12, 64, 32, device=dev)
    run(lambda: HF.conv3d_bwd_weight(gy, x, 2))
  else:
    x = torch.randn(2, 32, 48, 256, 128, device=dev)
    w = torch.randn(64, 32, 3, 3, 3, device=dev) * 0.05
    gy = torch.randn(2, 64, 24, 128, 64, device=dev)
    if what == 'conv3d_bwd_weight_s2':
      run(lambda: HF.conv3d_bwd_weight(gy, x, 2))
    elif what == 'conv3d_fwd_s2':
      run(lambda: HF.conv3d_fwd(x, w, 2))
    else:
      run(lambda: HF.conv3d_bwd_data(gy, w, x.shape, 2))
elif what in ('conv3d_co1_fwd', 'conv3d_co1_bwd_weight', 'conv3d_co1_bwd_data'):
  # the classifier heads' last layer: Conv3d(32 -> 1) at the full 48 x 256 x 128 volume, B = 2
  x = torch.randn(2, 32, 48, 256, 128, device=dev)
  w = torch.randn(1, 32, 3, 3, 3, device=dev) * 0.05
  gy = torch.randn(2, 1, 48, 256, 128, device=dev)
  if what == 'conv3d_co1_fwd':
    run(lambda: HF.conv3d_fwd(x, w, 1))
  elif what == 'conv3d_co1_bwd_weight':
    run(lambda: HF.conv3d_bwd_weight(gy, x, 1))
  else:
    run(lambda: HF.conv3d_bwd_data(gy, w, x.shape, 1))
elif what.startswith('conv3d') or what == 'bn3d_32':
  x = torch.randn(2, 32, 48, 256, 128, device=dev)
  w = torch.randn(32, 32, 3, 3, 3, device=dev) * 0.05
  gy = torch.randn_like(x)
  if what == 'bn3d_32':
    bn = torch.nn.BatchNorm3d(32).to(dev)
    xr = x.clone().requires_grad_(True)

    def step():
      xr.grad = None
      HF.bn_act(bn, xr, None, True).backward(gy)
    run(step)
  elif what == 'conv3d_bwd_weight_32':
    run(lambda: HF.conv3d_bwd_weight(gy, x, 1))
  elif what == 'conv3d_bwd_data_32':
    run(lambda: HF.conv3d_bwd_data(gy, w, x.shape, 1))
  else:
    run(lambda: HF.conv3d_fwd(x, w, 1))
elif what.startswith('sphere'):
  from models.basic.spherical_conv.sphere_conv import SphereConv
  m = SphereConv(256, 128, 'Cassini', 128, 128, 3, 1, 1).to(dev)
  pos = m.position_on(dev)
  H, W = pos.shape[2:]
  xt = torch.randn(4, 128, W, H, device=dev)
  gyt = torch.randn_like(xt)
  w = m.weight.detach()
  if what == 'sphere_fwd_t':
    yt = torch.empty_like(xt)
    run(lambda: HF.sphere_conv_fwd_t(xt, pos, w, yt, 1, f16=True))  # (the arithmetic of a training step)
  elif what == 'sphere_bwd_data_t':
    gxt = torch.empty_like(xt)
    run(lambda: HF.sphere_conv_bwd_data_t(gyt, pos, w, gxt, 1))
  else:
    gw = torch.zeros_like(w)
    run(lambda: HF.sphere_conv_bwd_weight_t(gyt, pos, xt, gw, 1))
elif what == 'cost_volume_fwd':
  fr = torch.randn(2, 32, 256, 128, device=dev)
  ft = torch.randn_like(fr)
  run(lambda: HF.cost_volume_fwd(fr, ft, 48))
else:
  raise SystemExit('unknown case ' + what)
print('done', what)
