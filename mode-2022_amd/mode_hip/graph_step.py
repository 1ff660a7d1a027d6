"""Replay a whole training (or inference) step as ONE hipGraph.

A ModeDisparity training step is ~2 200 kernel launches of 3 us .. 3 ms; launched one by one from Python the GPU idles for
about a quarter of the step.  The step has static shapes and no host-side data dependence (the sampling and adjoint tables
are constants of the geometry, the loss mask is applied arithmetically), so it is captured once -- forward, loss, backward
into the flat gradient buffer -- and replayed with a single launch.  The native library only ever enqueues on the stream it
is handed and never allocates, so capture needs nothing special from it.

The gradient exchange and the optimizer stay outside the graph: the exchange because a collective is only capturable with
RCCL (not with the gloo backend the CPU tests use), the optimizer so that its hyper-parameters stay ordinary Python state.

  gs = GraphedStep(fn, static_inputs)      # fn() reads the static input tensors and returns a tensor / tuple of tensors
  gs.load(left, right, gt, count)          # copy_ new data into the static inputs (optional)

EVERYTHING about a batch that fn() reads must be one of the static inputs -- also derived quantities such as a loss mask or the
global valid-pixel count: a tensor that fn() merely closes over is baked into the graph with the values (the address) it had at
capture time, and a replay on new data would silently use the stale one.  bench.py therefore passes the ground truth with its
NaNs (the mask is derived inside the step) and the valid-pixel count as static inputs.
  out = gs.replay()                        # static output tensors, overwritten by every replay

The graph's nodes hold the raw addresses of everything the kernels were launched with, including the per-geometry device tables of
mode_hip.functional (sampling tables, plans, adjoints).  Those caches are bounded LRUs; an entry they hand out while a capture is active
is pinned (functional._LRU) and never evicted, so a live graph cannot be left pointing at freed tables however many other geometries
the process touches afterwards.

Fixed-weights contract of captured INFERENCE.  An eval-mode forward keeps the packed (BatchNorm-folded) weights of every layer between
calls (functional._eval_wpack).  The warm-up above fills that cache, so the captured forward hits it and its graph contains NO pack
kernels: a replay computes with the packs of capture time.  That is what makes a replayed eval forward cheaper than an eager one
(0.65 ms of 10.3 at one pair), and it is only valid while the weights stand: the capture records (tensor, version) of everything the
hits vouched for, and replay() raises once any of them has been written by torch (optimizer step, load_state_dict, in-place op) or by
this library's training kernels -- launched eagerly or by a replayed training graph, whose replay() moves the version counters of the
BatchNorm state it writes; `stale()` tells without raising.  Writes through `.data` or foreign raw pointers move no version:
call functional.invalidate_eval_packs(model) and capture again after those.  A captured TRAINING step packs inside the graph and is
not affected.
"""
import torch

from . import functional


class GraphedStep(object):

  def __init__(self, fn, static_inputs=(), warmup=2, pool=None):
    self.static_inputs = list(static_inputs)
    dev = self.static_inputs[0].device if self.static_inputs else torch.device('cuda', torch.cuda.current_device())
    assert dev.type == 'cuda', 'hipGraph capture needs device tensors'
    # Warm up on a side stream: vendor find-phases, table/adjoint uploads, LDS attribute calls and allocator growth all
    # happen here, not during capture.
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
      for _ in range(warmup):
        fn()
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize(dev)
    self.graph = torch.cuda.CUDAGraph()
    functional.frozen_packs_begin()
    functional.written_log_begin()
    try:
      with torch.cuda.graph(self.graph, pool=pool):
        self.outputs = fn()
    finally:
      # (tensor, version) pairs whose packed copies the captured kernels read WITHOUT repacking (see the contract above)
      self.frozen = functional.frozen_packs_end()
      # tensors the captured kernels write through raw pointers (BatchNorm running statistics): every replay writes them again
      self.written = functional.written_log_end()
    torch.cuda.synchronize(dev)

  def load(self, *tensors):
    assert len(tensors) == len(self.static_inputs)
    for dst, src in zip(self.static_inputs, tensors):
      dst.copy_(src, non_blocking=True)

  def stale(self):
    """True when a weight / BatchNorm tensor whose PACKED copy the graph reads has been written since the capture."""
    return any(t._version != v for t, v in self.frozen)

  def replay(self):
    if self.frozen and self.stale():
      raise RuntimeError('GraphedStep: the weights of an eval-mode layer changed after this graph was captured; the graph reads the '
                         'packed copies of capture time -- capture a new GraphedStep (see the fixed-weights contract in graph_step.py)')
    self.graph.replay()
    if self.written:  # what the eager step does after every BatchNorm launch (functional._written_by_kernel): host-side only
      torch.autograd.graph.increment_version(self.written)
    return self.outputs
