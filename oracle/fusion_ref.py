"""CPU restatement of the fusion stage (TEST INFRASTRUCTURE ONLY -- imported by tests/ and nothing else).

A functional evaluation of ``ModeFusion.forward`` (models/mode_fusion.py:233-252, 301-313) from a state dict with plain
torch CPU ops, and the training loss of train_fusion.py:82-88, 99-112.  Pinned against the imported reference by
tests/golden/fusion_tiny.npz (tests/golden/make_golden_fusion.py; the reference file needs no stand-ins at all)."""
import torch
import torch.nn.functional as F


def _bn(P, key, x, train):
  return F.batch_norm(x, P[key + '.running_mean'], P[key + '.running_var'], P[key + '.weight'], P[key + '.bias'], train, 0.1, 1e-5)


def _block(P, key, x, train):
  """BasicBlock: conv1 = Sequential(Sequential(Conv2d, BN), ReLU), conv2 likewise (mode_fusion.py:21-34)."""
  for c in ('conv1', 'conv2'):
    x = F.relu(_bn(P, '%s.%s.0.1' % (key, c), F.conv2d(x, P['%s.%s.0.0.weight' % (key, c)], None, 1, 1), train))
  return x


def _layer(P, key, x, train, pool=False, blocks=1, up=False, head=False):
  i = 0
  if pool:
    x = F.max_pool2d(x, 2, 2)
    i += 1
  for _ in range(blocks):
    x = _block(P, '%s.%d' % (key, i), x, train)
    i += 1
  if up:
    x = F.conv_transpose2d(x, P['%s.%d.weight' % (key, i)], P['%s.%d.bias' % (key, i)], 2)
    x = F.relu(_bn(P, '%s.%d' % (key, i + 1), x, train))
  if head:
    x = torch.sigmoid(F.conv2d(x, P['%s.%d.weight' % (key, i)], P['%s.%d.bias' % (key, i)]))
  return x


def mode_fusion(P, depthes, confs, rgbs, maxdepth, train):
  fe = 'feature_extraction.'
  dc = []
  for d, c in zip(depthes, confs):
    dc += [d, c]
  depth_input, rgb_input = torch.cat(dc, 1), torch.cat(rgbs, 1)
  depth1 = _layer(P, fe + 'depth_layer1', depth_input, train, blocks=2)
  depth2 = _layer(P, fe + 'depth_layer2', depth1, train, pool=True)
  depth3 = _layer(P, fe + 'depth_layer3', depth2, train, pool=True)
  depth4 = _layer(P, fe + 'depth_layer4', depth3, train, pool=True, up=True)
  rgb1 = _layer(P, fe + 'rgb_layer1', rgb_input, train, blocks=2)
  rgb2 = _layer(P, fe + 'rgb_layer2', rgb1, train, pool=True)
  rgb3 = _layer(P, fe + 'rgb_layer3', rgb2, train, pool=True)
  fusion1 = _layer(P, fe + 'fusion_layer1', torch.cat((depth1, rgb1), 1), train, blocks=2)
  fusion2 = _layer(P, fe + 'fusion_layer2', torch.cat((depth2, rgb2), 1), train, blocks=2)
  fusion3 = _layer(P, fe + 'fusion_layer3', torch.cat((depth3, rgb3), 1), train, blocks=2)
  depth5 = _layer(P, fe + 'depth_layer5', torch.cat((fusion3, depth4), 1), train, up=True)
  depth6 = _layer(P, fe + 'depth_layer6', torch.cat((fusion2, depth5), 1), train, up=True)
  depth7 = _layer(P, fe + 'depth_layer7', torch.cat((fusion1, depth6), 1), train, blocks=2, head=True)
  return depth7 * maxdepth


def silog_loss(lamda, pred, gt):
  """train_fusion.py:82-88."""
  mask = (gt > 0) * (pred > 0)
  d = torch.log(pred[mask]) - torch.log(gt[mask])
  return torch.mean(torch.square(d)) - lamda * torch.square(torch.mean(d))


def training_loss(output, gt, maxdepth):
  """train_fusion.py:99-112: mask gt <= maxdepth, squeeze the channel, silog with lambda 0.5."""
  mask = gt <= maxdepth
  return silog_loss(0.5, torch.squeeze(output, 1)[mask], gt[mask])
