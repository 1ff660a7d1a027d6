"""Repro harness for the non-repeatable spherical input gradient under two processes sharing the GPU.

  python tools/experiments/adj9_race.py [--iters 3000] [--sync] [--H 32 --W 16] [--tag A]

Runs the NCHW operator chain of SphereConvFunction.backward's input gradient -- transpose_planes(gy) -> pack + adjoint gather kernel on
the transposed problem -> transpose_planes back -- `iters` times on fixed inputs, keeps every intermediate, and compares with the first
iteration AFTER the loop (no host sync inside it unless --sync).  Start two copies at the same time to get the contention."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'mode-2022_amd')):
  if p not in sys.path:
    sys.path.insert(0, p)

import torch  # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--iters', type=int, default=3000)
  ap.add_argument('--sync', action='store_true')
  ap.add_argument('--H', type=int, default=32)
  ap.add_argument('--W', type=int, default=16)
  ap.add_argument('--B', type=int, default=2)
  ap.add_argument('--C', type=int, default=128)
  ap.add_argument('--tag', default='A')
  ap.add_argument('--only', default='chain', choices=['chain', 'adj', 'transpose'])
  ap.add_argument('--seconds', type=float, default=0, help='keep running batches of --iters until this much time has passed')
  args = ap.parse_args()
  from mode_hip import functional as HF
  dev = torch.device('cuda', 0)
  torch.manual_seed(5)
  H, W, B, C = args.H, args.W, args.B, args.C
  from models.basic.spherical_conv.sphere_conv import make_sphere_position
  pos = make_sphere_position(min(H, W), max(H, W), 'Cassini', (3, 3)).to(dev).contiguous()  # Cassini image H x W (H = 2 W)
  assert tuple(pos.shape) == (1, 18, H, W), pos.shape
  w = torch.randn(C, C, 3, 3, device=dev) * 0.05
  gy = torch.randn(B, C, H, W, device=dev)

  def chain():
    gyt = HF.transpose_planes(gy)
    if args.sync:
      torch.cuda.synchronize()
    gxt = HF.sphere_conv_bwd_data_t(gyt, pos, w, torch.empty((B, C, W, H), device=dev), 1)
    if args.sync:
      torch.cuda.synchronize()
    gx = HF.transpose_planes(gxt, torch.empty((B, C, H, W), device=dev))
    return gyt, gxt, gx

  gyt_fixed = HF.transpose_planes(gy)

  def adj_only():
    gxt = HF.sphere_conv_bwd_data_t(gyt_fixed, pos, w, torch.empty((B, C, W, H), device=dev), 1)
    return (gxt,)

  def transpose_only():
    return (HF.transpose_planes(gy), HF.transpose_planes(gyt_fixed, torch.empty((B, C, H, W), device=dev)))

  fn = {'chain': chain, 'adj': adj_only, 'transpose': transpose_only}[args.only]
  names = {'chain': ('gyt', 'gxt', 'gx'), 'adj': ('gxt',), 'transpose': ('t1', 't2')}[args.only]
  ref = [t.clone() for t in fn()]
  torch.cuda.synchronize()
  total = bad = 0
  t0 = time.time()
  while True:
    outs = [fn() for _ in range(args.iters)]
    torch.cuda.synchronize()
    for it, o in enumerate(outs):
      for name, r, t in zip(names, ref, o):
        if not torch.equal(r, t):
          bad += 1
          ne = (r != t).nonzero()
          d = (r - t).abs()
          print('[%s] iter %d: %s differs: %d elements, max |d| %.3e (max |x| %.3e); first %s last %s; distinct (b) %s, channels %d, last-two-index box %s..%s' % (
              args.tag, total + it, name, ne.shape[0], float(d.max()), float(r.abs().max()), ne[0].tolist(), ne[-1].tolist(),
              sorted(set(ne[:, 0].tolist())), len(set(ne[:, 1].tolist())), ne[:, 2:].min(0).values.tolist(), ne[:, 2:].max(0).values.tolist()), flush=True)
          break
    total += args.iters
    del outs
    if time.time() - t0 >= args.seconds:
      break
  print('[%s] %s%s %dx%d: %d of %d iterations differ from the first (%.1f s)' % (args.tag, args.only, ' +sync' if args.sync else '', H, W, bad, total,
                                                                                  time.time() - t0), flush=True)


if __name__ == '__main__':
  main()
