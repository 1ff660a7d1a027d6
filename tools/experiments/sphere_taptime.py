"""Where a small-window workgroup of the spherical forward spends its cycles (debug build: MODE_HIP_DEFINES=MODE_TAPTIME python -m
mode_hip.build --force).  s_memtime of wave 0 at kernel entry, before the tap loop, after it, at the end; all tiles replaced by
small-window tiles so that every workgroup is stamped."""
import ctypes, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'mode-2022_amd')):
  sys.path.insert(0, p)
import numpy as np
import torch
from mode_hip import functional as HF
from models.basic.spherical_conv.sphere_conv import SphereConv
dev = torch.device('cuda', 0)
m = SphereConv(256, 128, 'Cassini', 128, 128, 3, 1, 1).to(dev)
pos = m.position_on(dev)
H, W = pos.shape[2:]
w = m.weight.detach()
HF.set_conv_arith('bf16x6')
tiles, (n0, n1, n2) = HF.sphere_plan(pos, 3, 3)[:2]
t = tiles.cpu().view(-1, 4).clone()
small = t[n1 + n2:n1 + n2 + n0]
for i in range(n1 + n2):
  t[i] = small[i % n0]
fake = t.view(-1).to(dev)
lib = HF.lib()
lib.mode_debug_taptime.argtypes = [ctypes.c_void_p, ctypes.c_int]
for B, use_fake in ((2, True), (4, True), (2, False)):
  xt = torch.randn(B, 128, W, H, device=dev)
  yt = torch.empty_like(xt)
  wp = torch.empty(lib.mode_sphere_conv_win_wpack_bytes(128, 128, 3, 3, 1) // 4, dtype=torch.float32, device=dev)
  for _ in range(3):
    if use_fake:
      HF._sphere_fwd_win(HF.ptr(xt), pos, w, None, HF.ptr(yt), wp, fake, n0 + n1 + n2, 0, 0, B, 128, H, W, 128, 3, 3, 1, 1, HF.stream_of(xt))
    else:
      HF._sphere_fwd_win(HF.ptr(xt), pos, w, None, HF.ptr(yt), wp, tiles, n0, n1, n2, B, 128, H, W, 128, 3, 3, 1, 1, HF.stream_of(xt))
  torch.cuda.synchronize()
  n = 128 * B
  dbuf = torch.zeros(8 * 8192, dtype=torch.int64, device=dev)
  assert lib.mode_debug_taptime(dbuf.data_ptr(), 8 * 8192) == 0
  out = dbuf.cpu().numpy()
  full = out[:8 * n].reshape(n, 8).astype(np.int64)
  s = full[:, :4]
  if not use_fake:
    tall = np.array([i for i in range(n) if (i % 128) < n1 + n2])
    for nm, sel in (('wrap-around', [i for i in tall if (i % 128) < n2]), ('145-row', [i for i in tall if (i % 128) >= n2])):
      q = s[sel]
      f = full[sel]
      print('   per step: first half %.0f, barrier %.0f, second half %.0f' % (f[:, 4].mean() / 80, f[:, 5].mean() / 80, f[:, 6].mean() / 80))
      print('real plan, %s tiles: prologue %.0f  pair-step loop %.0f = %.0f per step  epilogue %.0f' % (nm, (q[:, 1] - q[:, 0]).mean(), (q[:, 2] - q[:, 1]).mean(), (q[:, 2] - q[:, 1]).mean() / 80, (q[:, 3] - q[:, 2]).mean()))
    s = s[[i for i in range(n) if (i % 128) >= n1 + n2]]
    n = s.shape[0]
  pro, loop, epi = s[:, 1] - s[:, 0], s[:, 2] - s[:, 1], s[:, 3] - s[:, 2]
  sm = full[[i for i in range(full.shape[0]) if use_fake or (i % 128) >= n1 + n2]] if full.shape[0] != s.shape[0] else full
  print('   prologue of a small-window workgroup: entry -> zero fill + barrier %.0f, -> first window requested + records %.0f, -> window stored + barrier %.0f, -> two samples, first fragments %.0f' % ((sm[:, 4] - sm[:, 0]).mean(), (sm[:, 5] - sm[:, 4]).mean(), (sm[:, 6] - sm[:, 5]).mean(), (sm[:, 1] - sm[:, 6]).mean()))
  start = s[:, 0] - s[:, 0].min()
  print('%d images, %d workgroups: prologue %.0f (%.0f..%.0f)  tap loop %.0f = %.0f per tap (%.0f..%.0f)  epilogue %.0f (%.0f..%.0f) counter ticks; '
        'workgroup start %.0f..%.0f, end %.0f' % (B, n, pro.mean(), pro.min(), pro.max(), loop.mean(), loop.mean() / 72, loop.min() / 72, loop.max() / 72,
                                                 epi.mean(), epi.min(), epi.max(), start.min(), start.max(), (s[:, 3] - s[:, 0].min()).max()))
