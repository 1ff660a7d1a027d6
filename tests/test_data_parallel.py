"""CPU, world_size 2 over gloo: the gradient exchange of mode_hip.data_parallel reproduces the reference's
nn.DataParallel semantics (masked mean over the GLOBAL batch, gradients summed over replicas)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn
import torch.nn.functional as F

from mode_hip import data_parallel


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  return port


def _net():
  torch.manual_seed(3)
  return nn.Sequential(nn.Conv2d(3, 4, 3, padding=1), nn.ReLU(), nn.Conv2d(4, 1, 3, padding=1))


def _data():
  g = torch.Generator().manual_seed(5)
  x = torch.randn(4, 3, 8, 8, generator=g)
  gt = torch.randn(4, 1, 8, 8, generator=g)
  gt[0, 0, :6] = float('nan')  # very different valid-pixel counts on the two ranks
  gt[3, 0, 0, :2] = float('nan')
  return x, gt


def _worker(rank, world, port, out):
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  torch.set_num_threads(1)
  net = _net()
  if rank == 1:  # replicas start different; broadcast must fix that
    with torch.no_grad():
      for p in net.parameters():
        p.add_(1.0)
  red = data_parallel.GradAllReducer(net)
  red.broadcast_parameters(net)
  # one message per dtype (fp32 state, int64 counters), not one per tensor: a module with BatchNorm state
  def bn_net():
    torch.manual_seed(4)
    return nn.Sequential(nn.Conv2d(3, 4, 3), nn.BatchNorm2d(4), nn.Conv2d(4, 2, 1), nn.BatchNorm2d(2))
  m = bn_net()
  if rank == 1:
    with torch.no_grad():
      for t in list(m.parameters()) + list(m.buffers()):
        t.add_(3)
  r2 = data_parallel.GradAllReducer(m)
  r2.broadcast_parameters(m)
  assert r2.broadcast_messages == 2
  for a, b in zip(list(m.parameters()) + list(m.buffers()), list(bn_net().parameters()) + list(bn_net().buffers())):
    assert torch.equal(a, b) and a.dtype == b.dtype, rank
  x, gt = _data()
  xs, gts = x[rank * 2:rank * 2 + 2], gt[rank * 2:rank * 2 + 2]
  mask = ~torch.isnan(gts)
  for step in range(2):
    red.zero_grad()
    o = net(xs)
    loss = data_parallel.global_masked_mean(F.smooth_l1_loss(o, torch.nan_to_num(gts), reduction='none'), mask)
    loss.backward()
    red.all_reduce()
    with torch.no_grad():
      for p in net.parameters():
        p -= 0.1 * p.grad
  out[rank] = [p.detach().clone() for p in net.parameters()] + [red.flat.clone()]
  dist.destroy_process_group()


def test_two_ranks_match_single_process_global_batch():
  world = 2
  port = _free_port()
  mgr = mp.Manager()
  out = mgr.dict()
  mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
  # single-process reference: the whole batch, masked mean over all valid pixels (train_disparity.py:151-160)
  net = _net()
  x, gt = _data()
  mask = ~torch.isnan(gt)
  for step in range(2):
    net.zero_grad()
    o = net(x)
    loss = F.smooth_l1_loss(o[mask], gt[mask])
    loss.backward()
    with torch.no_grad():
      for p in net.parameters():
        p -= 0.1 * p.grad
  ref = [p.detach() for p in net.parameters()]
  for rank in range(world):
    got = out[rank]
    for a, b in zip(got[:-1], ref):
      assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)
  assert torch.equal(out[0][-1], out[1][-1])  # identical reduced gradient buffer on both ranks


def test_single_process_is_a_noop():
  net = _net()
  red = data_parallel.GradAllReducer(net)
  assert red.world == 1 and red.flat.numel() == sum(p.numel() for p in net.parameters())
  net(torch.randn(1, 3, 8, 8)).sum().backward()
  before = red.flat.clone()
  red.all_reduce()
  assert torch.equal(before, red.flat) and float(before.abs().sum()) > 0
  assert all(p.grad.data_ptr() >= red.flat.data_ptr() for p in net.parameters())
  net.zero_grad(set_to_none=True)
  red.rebind()
  assert all(p.grad is not None for p in net.parameters())


def test_bench_refuses_a_world_size_that_differs_from_gpus():
  """bench.py --gpus N either starts N ranks itself or runs inside a launcher that did; anything else is an error, never a
  silent 1-GPU run labelled N (checked before the GPU is touched, so this runs on the CPU tier)."""
  import subprocess
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
  r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'], capture_output=True,
                     text=True, timeout=300, env=env)
  assert r.returncode != 0 and '--gpus 2' in (r.stderr + r.stdout)


def test_bench_gpus_n_starts_n_ranks_itself():
  """Without WORLD_SIZE in the environment `bench.py --gpus 2` launches two ranks (torch.distributed.run child).  On this
  GPU-less host each rank stops at its "needs a GPU" assertion -- after the rendezvous, which is what is checked here."""
  import subprocess
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
  r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0', '--dist-backend', 'gloo',
                      '--no-cpu-baseline'], capture_output=True, text=True, timeout=600, env=env)
  out = r.stderr + r.stdout
  if torch.cuda.is_available():
    assert r.returncode == 0, out[-2000:]
  else:
    assert r.returncode != 0 and 'bench.py needs a GPU' in out and 'local_rank: 1' in out, out[-3000:]
