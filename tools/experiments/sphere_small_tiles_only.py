"""Windowed spherical forward with EVERY tile of the plan replaced by a small-window (class 0) tile: the time of the compact-tile code
alone (the results of the replaced tiles are wrong; the launch is the real one: 128 tiles per image).  Next to it the real plan."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'mode-2022_amd')):
  sys.path.insert(0, p)
import torch
from mode_hip import functional as HF
from models.basic.spherical_conv.sphere_conv import SphereConv
dev = torch.device('cuda', 0)
m = SphereConv(256, 128, 'Cassini', 128, 128, 3, 1, 1).to(dev)
pos = m.position_on(dev)
H, W = pos.shape[2:]
w = m.weight.detach()
HF.set_conv_arith('bf16x6')
plan = HF.sphere_plan(pos, 3, 3)
tiles, (n0, n1, n2) = plan[:2]
t = tiles.cpu().view(-1, 4).clone()
ntall = n1 + n2
small = t[ntall:ntall + n0]
fake = t.clone()
for i in range(ntall):
  fake[i] = small[i % n0]
fake = fake.view(-1).to(dev)


def run(tl, c0, c1, c2, B, what):
  xt = torch.randn(B, 128, W, H, device=dev)
  yt = torch.empty_like(xt)
  wp = torch.empty(HF.lib().mode_sphere_conv_win_wpack_bytes(128, 128, 3, 3, 1) // 4, dtype=torch.float32, device=dev)
  def call():
    HF._sphere_fwd_win(HF.ptr(xt), pos, w, None, HF.ptr(yt), wp, tl, c0, c1, c2, B, 128, H, W, 128, 3, 3, 1, 1, HF.stream_of(xt))
  for _ in range(3):
    call()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(20):
    call()
  e1.record()
  torch.cuda.synchronize()
  print('%-34s %d images: %.3f ms' % (what, B, e0.elapsed_time(e1) / 20))


for B in (2, 4, 8):
  run(tiles, n0, n1, n2, B, 'real plan (%d small, %d mid, %d wrap)' % (n0, n1, n2))
  run(fake, n0 + n1 + n2, 0, 0, B, 'all %d tiles small' % (n0 + n1 + n2))
