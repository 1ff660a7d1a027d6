"""ModeDisparity -- the disparity stage of MODE on MI355X.

Drop-in for the reference's models/mode_disparity.py: same constructor, ``forward(left, right)`` contract
(train: (pred1, pred2, pred3); eval: pred3, or (pred3, confidence) with out_conf=True), same module tree and
therefore the same 483 state_dict entries.  The forward pass itself is re-organised around the hand-written
gfx950 kernels of libmode_hip.so: spherical convolutions (inside the feature extractor), the single-pass cost
volume, and -- see ``stage3d`` -- the 3D regulariser and the fused soft-argmin head.
"""
from __future__ import print_function

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from mode_hip import functional as HF

from . import stage3d
from .submodule import convbn_3d, feature_extraction, sphere_feature_extraction


class hourglass(nn.Module):
  """Encoder-decoder 3D block (mode_disparity.py:11-46): two stride-2 convs down, two transposed convs up, with
  skip connections from an earlier hourglass (presqu/postsqu)."""

  def __init__(self, inplanes):
    super(hourglass, self).__init__()
    c2 = inplanes * 2
    self.conv1 = nn.Sequential(convbn_3d(inplanes, c2, kernel_size=3, stride=2, pad=1), nn.ReLU(inplace=True))
    self.conv2 = convbn_3d(c2, c2, kernel_size=3, stride=1, pad=1)
    self.conv3 = nn.Sequential(convbn_3d(c2, c2, kernel_size=3, stride=2, pad=1), nn.ReLU(inplace=True))
    self.conv4 = nn.Sequential(convbn_3d(c2, c2, kernel_size=3, stride=1, pad=1), nn.ReLU(inplace=True))
    self.conv5 = nn.Sequential(nn.ConvTranspose3d(c2, c2, kernel_size=3, padding=1, output_padding=1, stride=2, bias=False),
                               nn.BatchNorm3d(c2))
    self.conv6 = nn.Sequential(nn.ConvTranspose3d(c2, inplanes, kernel_size=3, padding=1, output_padding=1, stride=2, bias=False),
                               nn.BatchNorm3d(inplanes))

  def forward(self, x, presqu, postsqu, *, residual=None, pre_consumers=1, x_carrier=None):
    """Reference signature (x, presqu, postsqu).  The keyword-only `residual` is added to the output inside the last fused
    BatchNorm pass; the reference adds cost0 right after the call (mode_disparity.py:119, 122, 125).  `pre_consumers`: how many times
    the caller uses the returned `pre` (dres2's is the presqu of both later hourglasses): with more than one, `pre` comes back as a
    tuple of that many aliases whose gradients are summed together with the internal ones in one pass (functional.fan_out)."""
    # (x_carrier: x has one other consumer, HF.GradCarrier; None outside GPU training)
    out = stage3d.conv_bn(self.conv1[0], x, relu=True) if x_carrier is None else stage3d.conv_bn(self.conv1[0], x, relu=True, x_carrier=x_carrier)  # 1/4 -> 1/8
    pre = stage3d.conv_bn(self.conv2, out, relu=True, add=postsqu)  # relu(bn(conv) [+ postsqu])
    inner = 1 + (1 if presqu is None else 0)  # conv3, and conv5's skip when no presqu is given
    alias = HF.fan_out(pre, inner + pre_consumers) if inner + pre_consumers > 2 else (pre,) * (inner + pre_consumers)
    out = stage3d.conv_bn(self.conv3[0], alias[0], relu=True)  # 1/8 -> 1/16
    out = stage3d.conv_bn(self.conv4[0], out, relu=True)
    post = stage3d.conv_bn(self.conv5, out, relu=True, add=presqu if presqu is not None else alias[1])  # 1/16 -> 1/8
    out = stage3d.conv_bn(self.conv6, post, add=residual)  # 1/8 -> 1/4
    return out, (alias[inner] if pre_consumers == 1 else tuple(alias[inner:])), post


def _cumulative_bn(module):
  """True if some BatchNorm of `module` uses the cumulative moving average (momentum=None), which the grouped kernels do not do."""
  return any(isinstance(m, nn.modules.batchnorm._BatchNorm) and m.momentum is None for m in module.modules())


class ModeDisparity(nn.Module):
  """in_height, in_width: input image shape -- (1024,512) for Deep360, (640,320) fisheye, (512,256) 3D60.

  Two restructurings of forward() are plain attributes (defaults on; the parity tests switch them off per instance to compare
  against the literal composition -- they are not configuration and read no environment):
    pair_extractor     one pass of the shared extractor over [left; right] with per-image-set BatchNorm statistics;
    fold_cost_volume   the concat volume folded into dres0[0][0] (functional.cost_conv), never materialised."""
  pair_extractor = True
  fold_cost_volume = True

  def __init__(self, maxdisp, conv='Sphere', in_height=1024, in_width=512, sphereType='Cassini', out_conf=False):
    super(ModeDisparity, self).__init__()
    print("MODE stereo matching network!")
    self.maxdisp = maxdisp
    self.out_conf = out_conf
    if conv == 'Regular':
      self.feature_extraction = feature_extraction()
      print("using Regular feature extraction!")
    elif conv == 'Sphere':
      self.feature_extraction = sphere_feature_extraction(in_height, in_width, sphereType)
      print("using Spherical feature extraction!")
    else:
      raise NotImplementedError("Convolution Type must be Regular or Sphere!")

    relu = lambda: nn.ReLU(inplace=True)
    self.dres0 = nn.Sequential(convbn_3d(64, 32, 3, 1, 1), relu(), convbn_3d(32, 32, 3, 1, 1), relu())
    self.dres1 = nn.Sequential(convbn_3d(32, 32, 3, 1, 1), relu(), convbn_3d(32, 32, 3, 1, 1))
    self.dres2 = hourglass(32)
    self.dres3 = hourglass(32)
    self.dres4 = hourglass(32)
    for i in (1, 2, 3):
      setattr(self, 'classif%d' % i,
              nn.Sequential(convbn_3d(32, 32, 3, 1, 1), relu(), nn.Conv3d(32, 1, kernel_size=3, padding=1, stride=1, bias=False)))
    self._psmnet_init()

  def _psmnet_init(self):
    """mode_disparity.py:82-96: Conv2d/Conv3d ~ N(0, sqrt(2/(prod(kernel)*Cout))), BN gamma=1 beta=0; SphereConv and
    ConvTranspose3d keep their own default initialisation."""
    for m in self.modules():
      if isinstance(m, (nn.Conv2d, nn.Conv3d)):
        n = m.out_channels
        for k in m.kernel_size:
          n *= k
        m.weight.data.normal_(0, math.sqrt(2. / n))
      elif isinstance(m, (nn.BatchNorm2d, nn.BatchNorm3d)):
        m.weight.data.fill_(1)
        m.bias.data.zero_()
      elif isinstance(m, nn.Linear):
        m.bias.data.zero_()

  def forward(self, left, right):
    cost1, cost2, cost3 = self._logits(left, right)
    size = (self.maxdisp, left.size(2), left.size(3))
    if self.training:
      return stage3d.head(cost1, size), stage3d.head(cost2, size), stage3d.head(cost3, size)
    if self.out_conf:
      return stage3d.head(cost3, size, with_confidence=True)
    return stage3d.head(cost3, size)

  def forward_loss(self, left, right, disp_true, count=None, weights=(0.5, 0.7, 1.0)):
    """The body of the reference's training iteration between zero_grad() and backward() in one call (train_disparity.py:150-158):
        output1, output2, output3 = model(imgL, imgR)
        loss = 0.5 * smooth_l1(output1[mask], disp_true[mask]) + 0.7 * smooth_l1(output2[mask], ...) + smooth_l1(output3[mask], ...)
    with mask = the pixels that have a ground truth (NaN = none, train_disparity.py:195) and mean reduction over them.  An extension
    (the reference has no such method): returns (loss, (pred1, pred2, pred3)) -- the same numbers as forward() + those lines, with the
    loss and its gradient formed next to the soft-argmin heads (HF.head_loss) instead of by ~50 elementwise launches.  `count` = the number
    of valid pixels as a 0-d device tensor when it is known already (data_parallel.global_valid_count: the GLOBAL count when the batch is
    sharded over ranks); `disp_true` (B, 1, H, W) or (B, H, W) with NaN where there is no ground truth."""
    if not self.training:
      raise RuntimeError('forward_loss() is the training step; call the module itself for inference')
    costs = self._logits(left, right)
    size = (self.maxdisp, left.size(2), left.size(3))
    gt = disp_true.reshape(left.size(0), 1, left.size(2), left.size(3))
    if count is None:
      count = (~torch.isnan(gt)).sum().to(torch.float32).clamp(min=1)
    if HF.head_loss_supported(costs[0], size):
      return HF.head_loss(costs, size, gt, count.reciprocal(), weights)
    preds = tuple(stage3d.head(c, size) for c in costs)
    mask = ~torch.isnan(gt)
    gt0 = torch.nan_to_num(gt)
    loss = 0
    for wgt, o in zip(weights, preds):
      per = F.smooth_l1_loss(o, gt0, reduction='none')
      loss = loss + wgt * torch.where(mask, per, torch.zeros((), dtype=per.dtype, device=per.device)).sum() / count
    return loss, tuple(p.detach() for p in preds)

  def _logits(self, left, right):
    """Everything up to the three classifier outputs cost1, cost2, cost3 (B, 1, D/4, H/4, W/4) (mode_disparity.py:98-129)."""
    if left.is_cuda:
      with HF.weight_maxima(self):  # (training on the fp16 arithmetic: every weight's maximum from one launch; a no-op otherwise)
        return self._logits_body(left, right)
    return self._logits_body(left, right)

  def _logits_body(self, left, right):
    if self.pair_extractor and left.shape == right.shape and not _cumulative_bn(self.feature_extraction):
      # One pass of the shared extractor over [left; right] instead of two: same arithmetic per sample, BatchNorm statistics
      # still per image set (stage3d.bn_groups), twice the work per kernel launch -- the extractor's kernels are small at the
      # benchmark batch (2 x 256x128 at quarter resolution) and fill the chip better at 4.
      with stage3d.bn_groups(2):
        fea = self.feature_extraction(torch.cat((left, right), 0))
      ref_fea, tgt_fea = fea[:left.shape[0]], fea[left.shape[0]:]
    else:
      ref_fea = self.feature_extraction(left)
      tgt_fea = self.feature_extraction(right)

    conv0 = self.dres0[0][0]
    if (self.fold_cost_volume and ref_fea.is_cuda and conv0.bias is None and tuple(conv0.kernel_size) == (3, 3, 3) and
        tuple(conv0.stride) == (1, 1, 1) and tuple(conv0.padding) == (1, 1, 1) and conv0.in_channels == 2 * ref_fea.shape[1] and
        HF.cost_conv_supported(ref_fea, self.maxdisp // 4, conv0.out_channels)):
      # the volume is constant along d in its reference half and a function of w-d in its target half: its first convolution
      # collapses to 18 small 2-D products and one assembly pass (HF.cost_conv), and the 402.7 MB volume is never built
      if HF.bn_foldable(self.dres0[0][1]):  # inference: the BatchNorm + ReLU inside the assembly kernel
        cost0 = HF.cost_conv_bn_eval(ref_fea, tgt_fea, conv0.weight, self.maxdisp // 4, self.dres0[0][1], relu=True)
      else:
        y0 = HF.cost_conv(ref_fea, tgt_fea, conv0.weight, self.maxdisp // 4)
        cost0 = stage3d.bn_act(self.dres0[0][1], y0, None, True)
    else:
      cost = HF.cost_volume(ref_fea, tgt_fea, self.maxdisp // 4)  # (B, 64, D/4, H/4, W/4), one kernel
      cost0 = stage3d.conv_bn(self.dres0[0], cost, relu=True)
    cost0 = stage3d.conv_bn(self.dres0[2], cost0, relu=True)
    # Tensors with exactly two consumers -- dres1's input (its first convolution and its skip), out1 and out2 (a classifier head and the
    # next hourglass): the gradient that arrives first is added inside the kernel that produces the second (HF.GradCarrier) instead of
    # by a pass of autograd's over the 403 MB tensors.  The forward is unchanged.
    car = HF.grad_carrier(cost0)
    if car is None:
      t = stage3d.conv_bn(self.dres1[0], cost0, relu=True)
      cost0 = stage3d.conv_bn(self.dres1[2], t, add=cost0)
    else:
      t = stage3d.conv_bn(self.dres1[0], cost0, relu=True, x_carrier=car)
      cost0 = stage3d.conv_bn(self.dres1[2], t, add=cost0, add_carrier=car)

    # cost0 has four consumers (the input of dres2 and the three residual adds), pre1 four (two inside dres2, the presqu of dres3 and
    # dres4): their gradients are summed in one pass each instead of pairwise by autograd (HF.fan_out; the forward is unchanged)
    c0 = HF.fan_out(cost0, 4)
    out1, pre1, post1 = self.dres2(c0[0], None, None, residual=c0[1], pre_consumers=2)  # out1 = hourglass(...) + cost0
    car1 = HF.grad_carrier(out1)
    kw1 = {} if car1 is None else {'x_carrier': car1}
    out2, pre2, post2 = self.dres3(out1, pre1[0], post1, residual=c0[2], **kw1)
    car2 = HF.grad_carrier(out2)
    kw2 = {} if car2 is None else {'x_carrier': car2}
    out3, pre3, post3 = self.dres4(out2, pre1[1], post2, residual=c0[3], **kw2)  # pre1 (not pre2), as in the reference (:124)

    cost1 = stage3d.classify(self.classif1, out1, **kw1)
    cost2 = stage3d.classify(self.classif2, out2, add=cost1, **kw2)  # = classif2(out2) + cost1 (reference :128)
    cost3 = stage3d.classify(self.classif3, out3, add=cost2)
    return cost1, cost2, cost3
