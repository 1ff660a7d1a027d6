"""Batch-dimension data parallelism: one process per GPU, one RCCL all-reduce of the gradients per step.

Replaces the reference's single-process ``nn.DataParallel`` (train_disparity.py:264-265), which re-broadcasts the
21.96 MB of parameters every iteration, gathers the outputs on GPU 0 and reduces gradients there.  Semantics kept:
  * per-replica BatchNorm statistics (no SyncBN);
  * the loss is the masked mean over the GLOBAL batch (train_disparity.py:151-160 computes it once on the gathered
    outputs), so gradients are SUMMED over ranks and each rank scales its loss by 1 / global valid-pixel count
    (``global_masked_mean``) instead of averaging per-rank means.

All gradients live in ONE contiguous fp32 buffer (param.grad are views into it), so the exchange is a single
all-reduce of 5 489 280 floats with no flatten/unflatten copies; on the fully connected xGMI topology one large
message is the cheapest form (multi-GPU).  Works with the 'nccl' (= RCCL) and 'gloo' backends.
"""
import torch
import torch.distributed as dist


class GradAllReducer(object):

  def __init__(self, module, process_group=None, fuse_accumulation=True):
    """fuse_accumulation: let the native backward kernels add parameter gradients straight into the flat buffer
    (functional.grad_sink) instead of handing them to autograd's per-parameter accumulation kernels.  Needs zero_grad()
    of THIS object before every backward (optimizer.zero_grad(set_to_none=True) would detach the views; see rebind())."""
    self.group = process_group
    self.fuse_accumulation = fuse_accumulation
    self.params = [p for p in module.parameters() if p.requires_grad]
    n = sum(p.numel() for p in self.params)
    ref = self.params[0]
    self.flat = torch.zeros(n, dtype=ref.dtype, device=ref.device)
    off = 0
    for p in self.params:
      p.grad = self.flat[off:off + p.numel()].view_as(p)
      if fuse_accumulation:
        p._mode_grad_sink = p.grad
      off += p.numel()
    self.world = dist.get_world_size(self.group) if dist.is_initialized() else 1

  def broadcast_parameters(self, module, src=0):
    """Identical replicas at step 0 (DataParallel re-broadcasts every step; once is enough)."""
    if self.world == 1:
      return
    # ONE message per dtype instead of ~480 per-tensor broadcasts (83 weights + 80 x 5 BatchNorm entries; the 8-rank launch's start-up
    # was hundreds of small collectives): the state is packed into a flat buffer, broadcast, and copied back.  fp32 tensors -- all the
    # parameters and running statistics -- go as one 22 MB message, the int64 batch counters as a second, tiny one.
    by_dtype = {}
    for t in list(module.parameters()) + list(module.buffers()):
      by_dtype.setdefault((t.dtype, t.device), []).append(t.data)
    for (dtype, device), ts in by_dtype.items():
      flat = torch.cat([t.reshape(-1) for t in ts]) if len(ts) > 1 else ts[0].reshape(-1).clone()
      dist.broadcast(flat, src, group=self.group)
      off = 0
      for t in ts:
        t.copy_(flat[off:off + t.numel()].view_as(t))
        off += t.numel()
    self.broadcast_messages = len(by_dtype)

  def zero_grad(self):
    self.flat.zero_()

  def rebind(self):
    """Re-attach .grad views if something replaced them (e.g. optimizer.zero_grad(set_to_none=True))."""
    off = 0
    for p in self.params:
      if p.grad is None or p.grad.data_ptr() != self.flat.data_ptr() + off * self.flat.element_size():
        p.grad = self.flat[off:off + p.numel()].view_as(p)
      if self.fuse_accumulation:
        p._mode_grad_sink = p.grad
      off += p.numel()

  def detach(self):
    """Give the parameters back to plain autograd accumulation."""
    for p in self.params:
      if hasattr(p, '_mode_grad_sink'):
        del p._mode_grad_sink

  def all_reduce(self):
    """SUM over ranks (see module docstring for why not the mean)."""
    if self.world > 1:
      dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)


def global_valid_count(mask, group=None):
  """Number of valid pixels over ALL ranks as a 0-d fp32 device tensor (one small all-reduce when world > 1)."""
  count = mask.sum().to(torch.float32)
  if dist.is_initialized() and dist.get_world_size(group) > 1:
    dist.all_reduce(count, op=dist.ReduceOp.SUM, group=group)
  return count.clamp(min=1)


def global_masked_mean(per_pixel, mask, group=None, count=None):
  """sum(per_pixel[mask]) / (number of valid pixels over ALL ranks); differentiable w.r.t. per_pixel.  Pass ``count``
  (from global_valid_count, computed once per batch) to keep the collective out of the step -- required when the step is
  captured into a hipGraph."""
  if count is None:
    count = global_valid_count(mask, group)
  return torch.where(mask, per_pixel, torch.zeros((), dtype=per_pixel.dtype, device=per_pixel.device)).sum() / count
