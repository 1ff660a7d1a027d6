"""GPU (-m gpu): a two-rank data-parallel ModeDisparity step on the HIP path, numbers checked.

The reference's only parallelism is nn.DataParallel (train_disparity.py:264-265): replicas with their own BatchNorm statistics,
the loss a masked mean over the gathered (global) batch (train_disparity.py:151-161), gradients summed over the replicas.  The
replacement is one process per GPU (mode_hip/data_parallel.py).  An 8-GPU node is the driver's to run; what one GPU can prove is
proven here: two ranks (sharing the GPU, gloo backend -- RCCL refuses duplicate devices) run the step exactly as bench.py does --
gradient sinks (the native backward kernels add into the flat buffer), zero-grad + forward + loss + backward replayed as a hipGraph,
global masked mean, one all-reduce -- and the all-reduced flat gradient must equal
  (a) the sum of the two per-rank steps run in THIS process the plain way (autograd accumulation, eager launches), to fp32
      round-off, and
  (b) the gradient of the CPU oracle evaluated per replica with the global valid-pixel count, to the parity tier's whole-network
      bound (tests/test_gpu_parity.py: relative L2 <= 1e-3),
with per-replica BatchNorm running statistics and identical buffers on both ranks.  After this the only unverified thing about
`bench.py --gpus 8` is RCCL itself."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

import recipe
from oracle import mode_ref

import models
import mode_hip
import two_rank_worker as trw

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    return s.getsockname()[1]


def _run_ranks(tmp_path, maxdisp, H, W, launch):
  env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
  env['HSA_ENABLE_IPC_MODE_LEGACY'] = env.get('HSA_ENABLE_IPC_MODE_LEGACY', '0')
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port',
         str(_free_port()), os.path.join(ROOT, 'tests', 'two_rank_worker.py'), str(tmp_path), str(maxdisp), str(H), str(W), launch]
  r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env)
  assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
  return [torch.load(os.path.join(str(tmp_path), 'rank%d.pt' % k)) for k in range(2)]


def _rel(a, b):
  return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-300))


@pytest.mark.parametrize('maxdisp,H,W', [(32, 128, 64), (64, 512, 256)])
def test_two_rank_graph_step_equals_the_sum_of_single_process_steps(tmp_path, maxdisp, H, W):
  mode_hip.lib()
  ranks = _run_ranks(tmp_path, maxdisp, H, W, 'graph')
  assert all(r['launch'] == 'graph' for r in ranks)
  assert torch.equal(ranks[0]['flat'], ranks[1]['flat'])  # the same reduced buffer on both ranks
  assert torch.isfinite(ranks[0]['flat']).all() and float(ranks[0]['flat'].abs().sum()) > 0
  batches = [trw.rank_batch(k, maxdisp, H, W) for k in range(2)]
  count = float(sum((~torch.isnan(b[2])).sum() for b in batches))
  assert ranks[0]['count'] == count == ranks[1]['count']

  # (a) the same two steps, one after the other in this process: no sinks, no graph, no process group
  sd = recipe.recipe_state_wc(recipe.load_manifest(), 77)
  total, names = None, None
  for k, (left, right, gt) in enumerate(batches):
    net = models.ModeDisparity(maxdisp, 'Sphere', H, W, 'Cassini').to(DEV)
    net.load_state_dict(sd)
    net.train()
    loss = trw.step_loss(net, left.to(DEV), right.to(DEV), gt.to(DEV), torch.tensor(count, device=DEV))
    loss.backward()
    flat = torch.cat([p.grad.reshape(-1) for p in net.parameters() if p.requires_grad]).cpu()
    assert abs(float(loss) - ranks[k]['loss']) <= 1e-6 * abs(float(loss)) + 1e-9
    assert _rel(ranks[k]['local'], flat) <= 2e-5, ('rank %d local gradient (sinks + hipGraph) vs plain autograd' % k, _rel(ranks[k]['local'], flat))
    for key, v in ranks[k]['bn'].items():  # per-replica BatchNorm: rank k's running statistics are those of ITS sample alone
      assert torch.allclose(v, net.state_dict()[key].cpu(), rtol=1e-5, atol=1e-6), key
    total = flat if total is None else total + flat
    names = [n for n, p in net.named_parameters() if p.requires_grad]
  rel = _rel(ranks[0]['flat'], total)
  print('two ranks %dx%d/%d: |all-reduced flat gradient - sum of single-process steps| / |.| = %.2e' % (H, W, maxdisp, rel))
  assert rel <= 2e-5, rel
  assert not torch.allclose(ranks[0]['bn'][next(iter(ranks[0]['bn']))], ranks[1]['bn'][next(iter(ranks[1]['bn']))])  # (no SyncBN)

  # (b) the CPU oracle, replica by replica, global count (small size only: the oracle takes minutes at 512 x 256)
  if H * W > 128 * 64:
    return
  ref = None
  pos = mode_ref.sphere_position(H // 4, W // 4, 'Cassini')
  for left, right, gt in batches:
    P = {k: v.clone() for k, v in sd.items()}
    for k, v in P.items():
      if v.is_floating_point() and 'running' not in k:
        v.requires_grad_(True)
    preds = mode_ref.mode_disparity(P, left, right, maxdisp, pos, True)
    mask = ~torch.isnan(gt)
    gt0 = torch.nan_to_num(gt)
    loss = sum(w * torch.where(mask, torch.nn.functional.smooth_l1_loss(o, gt0, reduction='none'), torch.zeros(())).sum() / count
               for w, o in zip((0.5, 0.7, 1.0), preds))
    loss.backward()
    g = torch.cat([P[n].grad.reshape(-1) for n in names])
    ref = g if ref is None else ref + g
  rel = _rel(ranks[0]['flat'], ref)
  print('two ranks %dx%d/%d: against the CPU oracle (per-replica BatchNorm, global masked mean): %.2e' % (H, W, maxdisp, rel))
  assert rel <= 1e-3, rel


def _per_tensor_report(names, shapes, a, b, limit=10):
  """Which parameter tensors of two flat gradient buffers differ, and by how much (a bit-equality that fails must name the layer)."""
  lines, off = [], 0
  for name, shape in zip(names, shapes):
    n = int(np.prod(shape))
    x, y = a[off:off + n], b[off:off + n]
    off += n
    if not torch.equal(x, y):
      d = (x.double() - y.double()).abs()
      lines.append('%s %s: %d of %d elements differ, max |d| %.3e, rms of the tensor %.3e' % (
          name, tuple(shape), int((x != y).sum()), n, float(torch.nan_to_num(d).max()), float(x.double().pow(2).mean().sqrt())))
  assert off == a.numel() == b.numel()
  return '%d of %d parameter tensors differ' % (len(lines), len(names)) + ''.join('\n  ' + s for s in lines[:limit]) + \
      ('\n  ... (%d more)' % (len(lines) - limit) if len(lines) > limit else '')


def test_two_rank_eager_step_matches_the_graph_step(tmp_path):
  """Same step without the hipGraph: identical numbers (the graph is a launch optimisation only).  Every kernel of the step is
  deterministic by construction (fixed-order split-K, no atomics), so the bits must agree; when they do not, the message names the
  parameter tensors, per rank (`local` = a rank's own gradient before the all-reduce) and reduced (`flat`)."""
  a = [{k: (v.clone() if torch.is_tensor(v) else v) for k, v in r.items()} for r in _run_ranks(tmp_path, 32, 128, 64, 'eager')]
  b = _run_ranks(tmp_path, 32, 128, 64, 'graph')
  assert a[0]['launch'] == 'eager' and b[0]['launch'] == 'graph'
  names, shapes = a[0]['names'], a[0]['shapes']
  report = []
  for k in range(2):
    if not torch.equal(a[k]['local'], b[k]['local']):
      report.append('rank %d local gradient, eager vs graph (loss %.9g vs %.9g): %s' % (
          k, a[k]['loss'], b[k]['loss'], _per_tensor_report(names, shapes, a[k]['local'], b[k]['local'])))
  if not torch.equal(a[0]['flat'], b[0]['flat']):
    report.append('all-reduced gradient, eager vs graph: ' + _per_tensor_report(names, shapes, a[0]['flat'], b[0]['flat']))
  assert not report, '\n'.join(report)
  assert a[0]['loss'] == b[0]['loss'] and a[1]['loss'] == b[1]['loss']
