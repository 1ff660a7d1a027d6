"""Guard: inside `no_vendor_arithmetic()` any vendor-library arithmetic reached from the forward pass of the path raises.

Used by the tests (tests/no_vendor.py re-exports it) and by bench.py, which runs its first warm-up step under it so that the
timed step cannot silently be a vendor-library fallback.  The product has no backend switch, but a few host-side helpers still fall back to the torch module itself for shapes the HIP kernels do
not take (models/stage3d.conv3 -> `conv(x)`, bn_act_torch, Conv2d3x3Function beyond CONV2D_OWN_MAX_PIXELS).  Those exits are silent; this
guard makes them loud for the configurations that must stay on the hand-written kernels.  A torch-dispatch mode sees
every aten op of the calling thread: convolutions, BatchNorm, matrix products, softmax and interpolation are refused; elementwise glue
(cat, add, where, fills, the loss) is not arithmetic of the path and passes.  The backward pass needs no guard of its own: a vendor
backward kernel only exists for an op whose forward was the vendor's."""
import contextlib

import torch
from torch.utils._python_dispatch import TorchDispatchMode

FORBIDDEN = ('convolution', 'batch_norm', 'miopen', 'cudnn', 'aten.mm', 'aten.addmm', 'aten.bmm', 'aten.baddbmm', 'softmax', 'upsample_',
             'grid_sampler', 'aten.matmul', 'aten.linear')


class _Guard(TorchDispatchMode):

  def __init__(self):
    super().__init__()
    self.seen = 0

  def __torch_dispatch__(self, func, types, args=(), kwargs=None):
    name = str(func)
    if any(bad in name for bad in FORBIDDEN):
      raise AssertionError('vendor-library arithmetic on the path: %s' % name)
    self.seen += 1
    return func(*args, **(kwargs or {}))


@contextlib.contextmanager
def no_vendor_arithmetic():
  with _Guard() as g:
    yield g
