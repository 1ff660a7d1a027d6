"""One rank of a data-parallel ModeDisparity training step on the HIP path (started by tests/test_gpu_two_ranks.py through
`python -m torch.distributed.run`, the launcher the driver uses for bench.py): GradAllReducer(fuse_accumulation=True) -- the native
weight-gradient / BatchNorm backward kernels add straight into the flat gradient buffer --, zero-grad + forward + loss + backward
captured once and replayed as a hipGraph, loss = masked mean over the GLOBAL batch, one all-reduce of the flat buffer.  Ranks may
share one GPU (gloo backend; RCCL refuses duplicate devices).  Writes rank<r>.pt into the output directory.

    python -m torch.distributed.run --nproc-per-node 2 ... tests/two_rank_worker.py OUT_DIR MAXDISP H W [eager]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'mode-2022_amd'), os.path.join(ROOT, 'tests', 'golden')):
  if p not in sys.path:
    sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.nn.functional as F  # noqa: E402


def rank_batch(rank, maxdisp, H, W):
  """Per-rank pair + ground truth (different valid-pixel counts on the two ranks: the global masked mean must weigh them)."""
  import recipe
  left, right = recipe.recipe_images(1, H, W, 900 + rank, shift=4)
  gt = recipe.recipe_disparity_smooth(1, H, W, 910 + rank, maxdisp)
  if rank == 1:
    gt[:, :, :H // 3] = float('nan')
  return left, right, gt


def step_loss(net, left, right, gt, count):
  """0.5 / 0.7 / 1.0 smooth-L1 (train_disparity.py:151-160), summed over this rank's valid pixels and divided by the GLOBAL count."""
  from mode_hip import data_parallel
  mask = ~torch.isnan(gt)
  gt0 = torch.nan_to_num(gt)
  loss = 0
  for wgt, o in zip((0.5, 0.7, 1.0), net(left, right)):
    loss = loss + wgt * data_parallel.global_masked_mean(F.smooth_l1_loss(o, gt0, reduction='none'), mask, count=count)
  return loss


def main():
  out_dir, maxdisp, H, W = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
  launch = sys.argv[5] if len(sys.argv) > 5 else 'graph'
  rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
  dist.init_process_group('gloo', rank=rank, world_size=world)
  assert torch.cuda.is_available()
  torch.cuda.set_device(0)
  dev = torch.device('cuda', 0)
  import recipe
  import models
  from mode_hip import data_parallel
  from mode_hip.graph_step import GraphedStep

  net = models.ModeDisparity(maxdisp, 'Sphere', H, W, 'Cassini').to(dev)
  sd = recipe.recipe_state_wc(recipe.load_manifest(), 77)
  if rank == 1:  # replicas start different: broadcast_parameters must fix that
    sd = {k: (v + 0.5 if v.is_floating_point() and 'running' not in k else v) for k, v in sd.items()}
  net.load_state_dict(sd)
  net.train()
  reducer = data_parallel.GradAllReducer(net, fuse_accumulation=True)
  reducer.broadcast_parameters(net)
  left, right, gt = [t.to(dev) for t in rank_batch(rank, maxdisp, H, W)]
  count = data_parallel.global_valid_count(~torch.isnan(gt))

  def body():
    reducer.zero_grad()
    loss = step_loss(net, left, right, gt, count)
    loss.backward()
    return loss

  if launch == 'graph':
    # BatchNorm running statistics move with every warm-up / capture pass: restore them so that the ONE replay that counts starts
    # from the loaded state (the parent compares them per replica)
    bn0 = {k: v.clone() for k, v in net.state_dict().items() if 'running' in k or 'num_batches' in k}
    graphed = GraphedStep(body, (left, right, gt, count), warmup=1)
    for _ in range(3):  # the LAST of several replays is the one compared: a replay must not depend on what the previous one left behind
      with torch.no_grad():
        for k, v in net.state_dict().items():
          if k in bn0:
            v.copy_(bn0[k])
      reducer.flat.fill_(float('nan'))  # the replay itself must zero and fill the buffer
      loss = graphed.replay()
      torch.cuda.synchronize()
  else:
    loss = body()
  torch.cuda.synchronize()
  local = reducer.flat.clone()
  reducer.all_reduce()
  torch.cuda.synchronize()
  named = [(n, p) for n, p in net.named_parameters() if p.requires_grad]
  assert [p.data_ptr() for _, p in named] == [p.data_ptr() for p in reducer.params]
  torch.save({'flat': reducer.flat.cpu(), 'local': local.cpu(), 'loss': float(loss), 'count': float(count), 'launch': launch,
              'names': [n for n, _ in named], 'shapes': [tuple(p.shape) for _, p in named],
              'bn': {k: v.cpu() for k, v in net.state_dict().items() if 'running' in k}}, os.path.join(out_dir, 'rank%d.pt' % rank))
  dist.barrier()
  dist.destroy_process_group()


if __name__ == '__main__':
  main()
