"""Per-operator trace of a ModeDisparity step, for the repeatability tests (tests/test_gpu_repeat.py, tools/determinism_hunt.py).

Every autograd Function of the HIP path (mode_hip.functional + the spherical operator) gets its forward and backward wrapped: each
tensor a call returns is recorded -- a bit-level checksum (sum of the int32 view in int64: exact, order-free) and its NaN count, or,
with keep=True, a full clone.  Two runs of the same step on the same inputs must give the same trace; the FIRST differing entry (in
execution order) names the operator whose kernel is not repeatable, everything after it merely inherits the difference.  Clones can
also be recorded during a hipGraph capture (the copies become graph nodes; read them after a replay).

Weight gradients that go into gradient sinks are not returned by the Functions: compare the flat gradient buffer per parameter
(param_report)."""
import contextlib

import torch


def functions():
  from mode_hip import functional as HF
  from models.basic.spherical_conv import sphere_conv as SC
  out = []
  for mod in (HF, SC):
    for name in sorted(vars(mod)):
      obj = getattr(mod, name)
      if isinstance(obj, type) and issubclass(obj, torch.autograd.Function) and obj is not torch.autograd.Function and \
          obj.__module__ == mod.__name__:
        out.append(obj)
  return out


class Trace(object):

  def __init__(self, keep=False, inputs=False):
    self.keep = keep
    self.inputs = inputs  # also record the tensor arguments (and, for backward, the saved tensors) of every call
    self.labels = []
    self.items = []

  def record(self, label, t):
    if not (torch.is_tensor(t) and t.is_cuda and t.is_floating_point() and t.numel() > 0):
      return
    self.labels.append('%s %s' % (label, tuple(t.shape)))
    if self.keep:
      self.items.append(t.detach().clone())
    else:
      c = t.detach().contiguous()
      self.items.append(torch.stack((torch.sum(c.view(torch.int32), dtype=torch.int64), torch.isnan(c).sum())))

  def finish(self):
    """Host copy: keep=False -> (n, 2) int64 tensor; keep=True -> list of CPU tensors."""
    if self.keep:
      return [t.cpu() for t in self.items]
    return torch.stack(self.items).cpu() if self.items else torch.zeros((0, 2), dtype=torch.int64)


@contextlib.contextmanager
def tracing(trace):
  """Wrap forward / backward of every Function of the path while the context is active."""
  saved = []
  counters = {}

  def wrap(cls, which):
    orig = getattr(cls, which)

    def wrapped(ctx, *args):
      k = counters.get((cls.__name__, which), 0)
      counters[(cls.__name__, which)] = k + 1
      if trace.inputs:
        for i, t in enumerate(args):
          trace.record('%s.%s#%d.in[%d]' % (cls.__name__, which, k, i), t)
        if which == 'backward':
          for i, t in enumerate(ctx.saved_tensors):
            trace.record('%s.%s#%d.saved[%d]' % (cls.__name__, which, k, i), t)
      out = orig(ctx, *args)
      outs = out if isinstance(out, tuple) else (out,)
      for i, t in enumerate(outs):
        trace.record('%s.%s#%d[%d]' % (cls.__name__, which, k, i), t)
      return out

    saved.append((cls, which, cls.__dict__[which]))
    setattr(cls, which, staticmethod(wrapped))

  for cls in functions():
    wrap(cls, 'forward')
    wrap(cls, 'backward')
  try:
    yield trace
  finally:
    for cls, which, orig in saved:
      setattr(cls, which, orig)


def _is_input(label):
  return '.in[' in label or '.saved[' in label


def first_difference(labels, a, b):
  """(index, description) of the first differing OUTPUT entry of two finished traces, or None.  Recorded inputs (Trace(inputs=True))
  are there for the dump, not for the comparison: BatchNorm's running statistics are inputs that legitimately move with every step."""
  if isinstance(a, list):
    for i, (x, y) in enumerate(zip(a, b)):
      if _is_input(labels[i]):
        continue
      if not torch.equal(x, y) and not (torch.isnan(x) & torch.isnan(y)).all():
        d = (x.double() - y.double()).abs()
        bad = (x != y) & ~(torch.isnan(x) & torch.isnan(y))
        idx = bad.nonzero()
        return i, '%s: %d of %d elements differ, max |d| %.3e (max |x| %.3e), NaNs %d / %d, first at %s last at %s' % (
            labels[i], int(bad.sum()), x.numel(), float(torch.nan_to_num(d).max()), float(torch.nan_to_num(x).abs().max()),
            int(torch.isnan(x).sum()), int(torch.isnan(y).sum()), idx[0].tolist(), idx[-1].tolist())
    return None
  out = torch.tensor([not _is_input(lb) for lb in labels], dtype=torch.bool)
  ne = ((a != b).any(1) & out).nonzero()
  if ne.numel() == 0:
    return None
  i = int(ne[0])
  return i, '%s: checksum %d vs %d, NaNs %d vs %d (%d of %d entries differ)' % (labels[i], int(a[i, 0]), int(b[i, 0]), int(a[i, 1]), int(b[i, 1]),
                                                                                 ne.numel(), a.shape[0])


def nan_entries(labels, a):
  """Labels of the trace entries holding NaNs (an uninitialised workspace was read when the step ran with NaN-filled allocations)."""
  if isinstance(a, list):
    return [labels[i] for i, x in enumerate(a) if bool(torch.isnan(x).any())]
  return [labels[i] for i in range(a.shape[0]) if int(a[i, 1])]


def param_report(names_params, flat_a, flat_b, limit=12):
  """Per parameter tensor: differing elements / max |d| / rms of the reference, for two flat gradient buffers (CPU tensors)."""
  lines = []
  off = 0
  for name, p in names_params:
    n = p.numel()
    x, y = flat_a[off:off + n], flat_b[off:off + n]
    off += n
    if not torch.equal(x, y):
      d = (x.double() - y.double()).abs()
      lines.append('%s %s: %d of %d differ, max |d| %.3e, rms %.3e' % (name, tuple(p.shape), int((x != y).sum()), n, float(torch.nan_to_num(d).max()),
                                                                      float(x.double().pow(2).mean().sqrt())))
  head = '%d of %d parameter tensors differ' % (len(lines), len(names_params))
  return head + ''.join('\n  ' + s for s in lines[:limit]) + ('\n  ...' if len(lines) > limit else '')


@contextlib.contextmanager
def nan_filled_allocations():
  """torch.empty & co. return NaN-filled memory inside the context (torch.utils.deterministic.fill_uninitialized_memory): a kernel
  that reads a workspace slot nobody wrote turns its output into NaN instead of reading whatever the allocator had there."""
  import torch.utils.deterministic as D
  prev = (torch.are_deterministic_algorithms_enabled(), torch.is_deterministic_algorithms_warn_only_enabled(), D.fill_uninitialized_memory)
  torch.use_deterministic_algorithms(True, warn_only=True)
  D.fill_uninitialized_memory = True
  try:
    yield
  finally:
    torch.use_deterministic_algorithms(prev[0], warn_only=prev[1])
    D.fill_uninitialized_memory = prev[2]
