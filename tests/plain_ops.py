"""Plain-torch compositions of layers the product runs as fused HIP kernels.  TEST / MEASUREMENT INFRASTRUCTURE: used by the
CPU wiring tests (tests/test_host.py, tests/test_fusion.py) and by the A/B timings in tools/; never imported by the product."""
import torch
import torch.nn.functional as F


def head(cost, size, with_confidence=False):
  """mode_disparity.py:131-152 (+ :157-183): trilinear upsample, softmax over D, expectation; optional confidence map."""
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
