"""Functional CPU restatement of the reference's ``ModeDisparity`` forward.  TEST INFRASTRUCTURE.

The network is restated as pure functions over a flat ``{state_dict key: tensor}`` mapping
(no nn.Module tree), using torch's CPU kernels for the stock layers and
``oracle.sphere_conv_ref`` for the custom op.  Autograd through these functions gives the
reference gradients.  Reference (paths relative to the upstream repo):

  models/mode_disparity.py:98-185   ModeDisparity.forward           -> ``mode_disparity``
  models/mode_disparity.py:104-113  concat cost volume              -> ``cost_volume``
  models/mode_disparity.py:11-46    hourglass                       -> ``hourglass``
  models/mode_disparity.py:131-152, models/submodule.py:50-57  head -> ``disparity_head``
  models/mode_disparity.py:157-183  confidence map                  -> ``confidence_map``
  models/submodule.py:151-201       sphere_feature_extraction       -> ``sphere_features``
  models/submodule.py:15-22, 94-147 convbn / convbn_3d / basic blocks
  models/basic/spherical_conv/sphere_conv.py:180-237  sampling table -> ``sphere_position``

Pinned against the imported reference by tests/golden/*.npz (made by
tests/golden/make_golden.py); see oracle/__init__.py for what is and is not pinned.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import sphere_conv_ref

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


# ----------------------------------------------------------------------------- table
def sphere_position(in_height, in_width, sphere_type, kernel_size=(3, 3)):
  """SphereConv.gen_sphere_position, sphere_conv.py:180-237, vectorised over rows.

  Every float64 operation of the reference is kept in its original order (so the float32
  table is bit-identical); only the Python-level list comprehensions over rows/columns are
  replaced by broadcasting."""
  h_ = min(in_height, in_width)
  w_ = max(in_height, in_width)
  assert w_ == 2 * h_
  height, width = h_, w_
  Kh, Kw = kernel_size
  delta_lat = np.pi / height
  delta_lon = 2 * np.pi / width
  range_x = np.arange(-(Kw // 2), Kw // 2 + 1)
  if not Kw % 2:
    range_x = np.delete(range_x, Kw // 2)
  range_y = np.arange(-(Kh // 2), Kh // 2 + 1)
  if not Kh % 2:
    range_y = np.delete(range_y, Kh // 2)
  kerX = np.tan(range_x * delta_lon)
  kerY = np.tan(range_y * delta_lat) / np.cos(range_y * delta_lon)  # sic: delta_lon (:194)
  kerX, kerY = np.meshgrid(kerX, kerY)
  rho = np.sqrt(kerX**2 + kerY**2)
  if Kh % 2 and Kw % 2:
    rho[Kh // 2][Kw // 2] = 1e-8
  nu = np.arctan(rho)
  cos_nu = np.cos(nu)
  sin_nu = np.sin(nu)
  lat_range = ((np.arange(0, height) / height) - 0.5) * np.pi
  lon_range = ((np.arange(0, width) / width) - 0.5) * (2 * np.pi)
  sl = np.sin(lat_range)[:, None, None]
  cl = np.cos(lat_range)[:, None, None]
  lat = np.arcsin(cos_nu * sl + kerY * sin_nu * cl / rho)  # (H,Kh,Kw)
  lon = np.arctan2(kerX * sin_nu, (rho * cl * cos_nu - kerY * sl * sin_nu))  # (H,Kh,Kw)
  lat = np.broadcast_to(lat[:, None], (height, width, Kh, Kw))
  lon = lon[:, None] + lon_range[None, :, None, None]
  lat = (lat / np.pi + 0.5) * height
  lon = ((lon / (2 * np.pi) + 0.5) * width) % width
  if sphere_type == 'ERP':
    t = np.stack((lat, lon)).astype(np.float32).transpose((3, 4, 0, 1, 2))
  elif sphere_type == 'Cassini':
    t = np.stack((lon, lat)).astype(np.float32).transpose((3, 4, 0, 2, 1))
  else:
    raise ValueError(sphere_type)
  Kh_, Kw_, d, H, W = t.shape
  return torch.from_numpy(np.ascontiguousarray(t.reshape((1, d * Kh_ * Kw_, H, W))))


# ----------------------------------------------------------------------------- layers
def _bn(P, name, x, train):
  return F.batch_norm(x, P[name + '.running_mean'], P[name + '.running_var'], P[name + '.weight'],
                      P[name + '.bias'], training=train, momentum=BN_MOMENTUM, eps=BN_EPS)


def _convbn2d(P, name, x, train, stride=1, pad=0, dil=1):
  """convbn, submodule.py:15-17: padding = dilation if dilation > 1 else pad."""
  y = F.conv2d(x, P[name + '.0.weight'], None, stride, dil if dil > 1 else pad, dil)
  return _bn(P, name + '.1', y, train)


def _convbn3d(P, name, x, train, stride=1):
  """convbn_3d, submodule.py:20-22 (always k3 p1 here)."""
  return _bn(P, name + '.1', F.conv3d(x, P[name + '.0.weight'], None, stride, 1), train)


def _sphere_convbn(P, name, x, pos, train):
  """sphereConvbn, submodule.py:61-74 (stride 1, pad 1 at every call site)."""
  y = sphere_conv_ref.sphere_conv(x, pos, P[name + '.0.weight'], None, 1, 1, 1, 1)
  return _bn(P, name + '.1', y, train)


def _regular_block(P, name, x, train, stride, dil):
  """RegularBasicBlock.forward, submodule.py:109-119."""
  out = F.relu(_convbn2d(P, name + '.conv1.0', x, train, stride, 1, dil))
  out = _convbn2d(P, name + '.conv2', out, train, 1, 1, dil)
  if (name + '.downsample.0.weight') in P:
    x = _bn(P, name + '.downsample.1', F.conv2d(x, P[name + '.downsample.0.weight'], None, stride), train)
  return F.relu(out + x)


def _sphere_block(P, name, x, pos, train):
  """SphereBasicBlock.forward, submodule.py:136-147."""
  out = F.relu(_sphere_convbn(P, name + '.conv1.0', x, pos, train))
  out = _sphere_convbn(P, name + '.conv2', out, pos, train)
  if (name + '.downsample.0.weight') in P:
    x = _bn(P, name + '.downsample.1', F.conv2d(x, P[name + '.downsample.0.weight']), train)
  return F.relu(out + x)


def sphere_features(P, x, pos, train, prefix='feature_extraction'):
  """sphere_feature_extraction.forward, submodule.py:192-201 (layer config :155-162)."""
  p = prefix
  y = F.relu(_convbn2d(P, p + '.firstconv.0', x, train, 2, 3, 1))
  y = F.relu(_convbn2d(P, p + '.firstconv.2', y, train, 1, 1, 1))
  y = F.relu(_convbn2d(P, p + '.firstconv.4', y, train, 1, 1, 1))
  for i in range(3):
    y = _regular_block(P, '%s.layer1.%d' % (p, i), y, train, 1, 1)
  for i in range(8):
    y = _regular_block(P, '%s.layer2.%d' % (p, i), y, train, 2 if i == 0 else 1, 1)
  raw = y
  for i in range(4):
    y = _regular_block(P, '%s.layer3.%d' % (p, i), y, train, 1, 2)
  regular = y
  for i in range(8):
    y = _sphere_block(P, '%s.layer4.%d' % (p, i), y, pos, train)
  f = torch.cat((raw, regular, y), 1)
  f = F.relu(_convbn2d(P, p + '.lastconv.0', f, train, 1, 0, 1))
  f = F.relu(_convbn2d(P, p + '.lastconv.2', f, train, 1, 1, 1))
  f = F.relu(_convbn2d(P, p + '.lastconv.4', f, train, 1, 0, 1))
  return f


def cost_volume(ref, tgt, d4):
  """mode_disparity.py:104-113: cost[b,c,i,h,w] = ref[b,c,h,w], cost[b,C+c,i,h,w] = tgt[b,c,h,w-i]
  for w >= i, zero elsewhere."""
  B, C, H, W = ref.shape
  cost = ref.new_zeros((B, 2 * C, d4, H, W))
  for i in range(d4):
    if i > 0:
      cost[:, :C, i, :, i:] = ref[:, :, :, i:]
      cost[:, C:, i, :, i:] = tgt[:, :, :, :-i]
    else:
      cost[:, :C, i] = ref
      cost[:, C:, i] = tgt
  return cost


def _deconvbn3d(P, name, x, train):
  """ConvTranspose3d k3 s2 p1 op1 + BatchNorm3d, mode_disparity.py:23, 25."""
  y = F.conv_transpose3d(x, P[name + '.0.weight'], None, 2, 1, 1)
  return _bn(P, name + '.1', y, train)


def hourglass(P, name, x, presqu, postsqu, train):
  """hourglass.forward, mode_disparity.py:27-46."""
  out = F.relu(_convbn3d(P, name + '.conv1.0', x, train, 2))
  pre = _convbn3d(P, name + '.conv2', out, train)
  pre = F.relu(pre + postsqu) if postsqu is not None else F.relu(pre)
  out = F.relu(_convbn3d(P, name + '.conv3.0', pre, train, 2))
  out = F.relu(_convbn3d(P, name + '.conv4.0', out, train))
  c5 = _deconvbn3d(P, name + '.conv5', out, train)
  post = F.relu(c5 + presqu) if presqu is not None else F.relu(c5 + pre)
  out = _deconvbn3d(P, name + '.conv6', post, train)
  return out, pre, post


def _classif(P, name, x, train):
  """classif1-3, mode_disparity.py:76-80."""
  y = F.relu(_convbn3d(P, name + '.0', x, train))
  return F.conv3d(y, P[name + '.2.weight'], None, 1, 1)


def disparity_head(cost, maxdisp, H, W, return_prob=False):
  """mode_disparity.py:131-152 + submodule.py:50-57: trilinear upsample (align_corners=True) to
  (maxdisp,H,W), softmax over disparity, expectation of d = 0..maxdisp-1."""
  up = F.interpolate(cost, [maxdisp, H, W], mode='trilinear', align_corners=True).squeeze(1)
  prob = F.softmax(up, dim=1)
  disp = torch.arange(maxdisp, dtype=cost.dtype).view(1, maxdisp, 1, 1)
  pred = torch.sum(prob * disp, 1, keepdim=True)
  return (pred, prob) if return_prob else pred


def confidence_map(pred, prob):
  """mode_disparity.py:157-183: sum of the probabilities at round(d)-1, round(d), round(d)+1, each
  looked up with nearest sampling and border clamping along the disparity axis."""
  D = prob.shape[1]
  r = torch.round(pred)
  out = 0
  for off in (0.0, -1.0, 1.0):
    # grid_sample(nearest, align_corners=True, border): index = round(clamp(x, 0, D-1)) where the
    # normalised coordinate maps back to x = r + off exactly up to float rounding.
    g = (r + off) / (D - 1.0) * 2 - 1
    x = ((g + 1) / 2) * (D - 1)
    idx = torch.round(x.clamp(0, D - 1)).long()  # torch.round = nearbyint = half-to-even
    out = out + torch.gather(prob, 1, idx)
  return out


def mode_disparity(P, left, right, maxdisp, pos, train, out_conf=False, taps=None):
  """ModeDisparity.forward, mode_disparity.py:98-185.  ``P`` maps state_dict keys to tensors
  (running statistics are updated in place in train mode, as nn.BatchNorm does).  ``taps``, if a
  dict, receives named intermediates."""
  fl = sphere_features(P, left, pos, train)
  fr = sphere_features(P, right, pos, train)
  cost = cost_volume(fl, fr, maxdisp // 4)
  c0 = F.relu(_convbn3d(P, 'dres0.0', cost, train))
  c0 = F.relu(_convbn3d(P, 'dres0.2', c0, train))
  c1 = F.relu(_convbn3d(P, 'dres1.0', c0, train))
  cost0 = _convbn3d(P, 'dres1.2', c1, train) + c0
  out1, pre1, post1 = hourglass(P, 'dres2', cost0, None, None, train)
  out1 = out1 + cost0
  out2, pre2, post2 = hourglass(P, 'dres3', out1, pre1, post1, train)
  out2 = out2 + cost0
  out3, pre3, post3 = hourglass(P, 'dres4', out2, pre1, post2, train)  # pre1 again (:124)
  out3 = out3 + cost0
  cost1 = _classif(P, 'classif1', out1, train)
  cost2 = _classif(P, 'classif2', out2, train) + cost1
  raw3 = _classif(P, 'classif3', out3, train)
  cost3 = raw3 + cost2
  H, W = left.shape[2:]
  if taps is not None:
    taps.update(fea_left=fl, fea_right=fr, cost=cost, cost0=cost0, out1=out1, out3=out3,
                logits1=cost1, logits2=cost2, logits3=cost3, classif3_raw=raw3)
  if train:
    return (disparity_head(cost1, maxdisp, H, W), disparity_head(cost2, maxdisp, H, W),
            disparity_head(cost3, maxdisp, H, W))
  pred3, prob = disparity_head(cost3, maxdisp, H, W, return_prob=True)
  if out_conf:
    return pred3, confidence_map(pred3, prob)
  return pred3


def training_loss(preds, disp_true, mask):
  """train_disparity.py:152-158 (size_average=True == mean)."""
  o1, o2, o3 = preds
  return (0.5 * F.smooth_l1_loss(o1[mask], disp_true[mask]) + 0.7 * F.smooth_l1_loss(o2[mask], disp_true[mask]) +
          F.smooth_l1_loss(o3[mask], disp_true[mask]))
