"""3D regulariser + soft-argmin head building blocks used by ModeDisparity.forward.

Each helper takes the nn.Module that owns the parameters (so the state_dict layout stays the reference's) and
runs the layer.  ``BACKEND`` selects who does the arithmetic for the stock 3D layers:
  'vendor' -- torch.nn.functional on the GPU (MIOpen);
  'hip'    -- libmode_hip.so kernels (as they land; see DESIGN.md for the current coverage).
Neither is a CPU path.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

BACKEND = os.environ.get('MODE_STAGE3D', 'vendor')


def conv_bn(seq, x, relu=False, add=None):
  """seq = Sequential(Conv3d | ConvTranspose3d, BatchNorm3d): y = bn(conv(x)) [+ add] [relu]
  (convbn_3d submodule.py:20-22; transposed form mode_disparity.py:23, 25)."""
  y = seq[1](seq[0](x))
  if add is not None:
    y = y + add
  return F.relu(y, inplace=True) if relu else y


def classify(seq, x):
  """classifN = Sequential(convbn_3d, ReLU, Conv3d(32->1)) (mode_disparity.py:76-80)."""
  return seq[2](conv_bn(seq[0], x, relu=True))


def head(cost, size, with_confidence=False):
  """Trilinear upsample (align_corners=True) of (B,1,D/4,H/4,W/4) logits to `size`=(D,H,W), softmax over D and
  expectation of the disparity index (mode_disparity.py:131-152, submodule.py:50-57); optionally the confidence
  map of mode_disparity.py:157-183 = P(round(d)-1) + P(round(d)) + P(round(d)+1), indices clamped to [0, D-1]."""
  D = size[0]
  up = F.interpolate(cost, list(size), mode='trilinear', align_corners=True).squeeze(1)
  prob = F.softmax(up, dim=1)
  disp = torch.arange(D, dtype=prob.dtype, device=prob.device).view(1, D, 1, 1)
  pred = torch.sum(prob * disp, 1, keepdim=True)
  if not with_confidence:
    return pred
  r = torch.round(pred)
  conf = 0
  for off in (0.0, -1.0, 1.0):
    idx = (r + off).clamp(0, D - 1).long()
    conf = conf + torch.gather(prob, 1, idx)
  return pred, conf  # (B,1,H,W) each, like the reference's prob_map.squeeze(1)
