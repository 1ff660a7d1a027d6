"""fp64 CPU restatement of the stock 3x3x3 layers of the regulariser at FULL benchmark size.  TEST INFRASTRUCTURE ONLY.

The reference runs these layers as ``nn.Conv3d`` / ``nn.ConvTranspose3d`` (convbn_3d, models/submodule.py:20-22; hourglass,
models/mode_disparity.py:11-46; classifiers :76-80) -- i.e. cuDNN / torch.  torch's own CPU conv3d in fp64 needs minutes at
48 x 256 x 128, so the full-size GPU tests (tests/test_gpu_fullsize.py) use this tap-wise form instead: one (Co x Ci) @
(Ci x P) float64 GEMM per kernel tap on a strided view of the zero-padded input -- 27 GEMMs, a few seconds per layer.

Pinned against torch's CPU convolutions (what oracle/mode_ref.py, and through it the imported reference, computes with) in
tests/test_oracle_golden.py::test_tapwise_conv_oracle_*.

    y[b, o, q] = sum_{c, k} w[o, c, k] * x[b, c, s*q + k - 1]          (k3, padding 1, stride s, no bias)
"""
import torch


def _pad1(x):
  return torch.nn.functional.pad(x, (1, 1, 1, 1, 1, 1))


def _out(n, s):
  return (n - 1) // s + 1


def _tap_view(xp, kd, kh, kw, s, Do, Ho, Wo):
  """x_padded[..., s*q + k] for all output positions q, as a strided view (B, C, Do, Ho, Wo)."""
  return xp[:, :, kd:kd + s * (Do - 1) + 1:s, kh:kh + s * (Ho - 1) + 1:s, kw:kw + s * (Wo - 1) + 1:s]


def conv3d_fwd(x, w, stride=1):
  """F.conv3d(x, w, None, stride, 1) for a (Co, Ci, 3, 3, 3) weight, float64."""
  x, w = x.double(), w.double()
  B, Ci, D, H, W = x.shape
  Co = w.shape[0]
  Do, Ho, Wo = _out(D, stride), _out(H, stride), _out(W, stride)
  xp = _pad1(x)
  y = torch.zeros((B, Co, Do * Ho * Wo), dtype=torch.float64)
  for kd in range(3):
    for kh in range(3):
      for kw in range(3):
        xs = _tap_view(xp, kd, kh, kw, stride, Do, Ho, Wo).reshape(B, Ci, -1)
        y += torch.matmul(w[:, :, kd, kh, kw], xs)
  return y.view(B, Co, Do, Ho, Wo)


def conv3d_bwd_weight(gy, x, stride=1):
  """Weight gradient of conv3d_fwd: gW[o, c, k] = sum_{b, q} gy[b, o, q] * x[b, c, s*q + k - 1]."""
  gy, x = gy.double(), x.double()
  B, Ci, D, H, W = x.shape
  Co, Do, Ho, Wo = gy.shape[1:]
  xp = _pad1(x)
  g2 = gy.reshape(B, Co, -1)
  gw = torch.zeros((Co, Ci, 3, 3, 3), dtype=torch.float64)
  for kd in range(3):
    for kh in range(3):
      for kw in range(3):
        xs = _tap_view(xp, kd, kh, kw, stride, Do, Ho, Wo).reshape(B, Ci, -1)
        gw[:, :, kd, kh, kw] = torch.einsum('bop,bcp->oc', g2, xs)
  return gw


def conv3d_bwd_data(gy, w, in_shape, stride=1):
  """Input gradient of conv3d_fwd: gx[b, c, s*q + k - 1] += sum_o w[o, c, k] * gy[b, o, q]."""
  gy, w = gy.double(), w.double()
  B, Ci, D, H, W = in_shape
  Co, Do, Ho, Wo = gy.shape[1:]
  gxp = torch.zeros((B, Ci, D + 2, H + 2, W + 2), dtype=torch.float64)
  g2 = gy.reshape(B, Co, -1)
  for kd in range(3):
    for kh in range(3):
      for kw in range(3):
        t = torch.matmul(w[:, :, kd, kh, kw].t(), g2).view(B, Ci, Do, Ho, Wo)
        _tap_view(gxp, kd, kh, kw, stride, Do, Ho, Wo).add_(t)
  return gxp[:, :, 1:D + 1, 1:H + 1, 1:W + 1].contiguous()


def deconv3d_fwd(x, w):
  """F.conv_transpose3d(x, w, None, 2, 1, 1) for a (Cin, Cout, 3, 3, 3) weight (mode_disparity.py:23, 25): the input gradient
  of the stride-2 convolution that has `w` as its (Co = Cin, Ci = Cout) weight, on a (2D, 2H, 2W) volume."""
  B, Cin, D, H, W = x.shape
  return conv3d_bwd_data(x, w, (B, w.shape[1], 2 * D, 2 * H, 2 * W), 2)
