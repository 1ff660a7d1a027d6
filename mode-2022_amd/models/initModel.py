"""Model initialisation / partial checkpoint loading (API of the reference's models/initModel.py)."""
import torch
import torch.nn as nn

from .basic import SphereConv

_CONV_TYPES = (nn.Conv2d, nn.Conv3d, nn.ConvTranspose2d, nn.ConvTranspose3d, SphereConv)
_WEIGHT_INIT = {
    'kaiming_normal': lambda w: nn.init.kaiming_normal_(w, mode='fan_in', nonlinearity='leaky_relu'),
    'xavier_normal': nn.init.xavier_normal_,
    'kaiming_uniform': lambda w: nn.init.kaiming_uniform_(w, mode='fan_in', nonlinearity='leaky_relu'),
    'xavier_uniform': nn.init.xavier_uniform_,
    'normal': nn.init.normal_,
}


def _zero_bias(module):
  if getattr(module, 'bias', None) is not None:
    nn.init.constant_(module.bias, 0)


def initModelPara(model, initType):
  """Re-draws the weights of every convolution-like layer with the named scheme (unknown names leave them alone), sets
  BatchNorm1d/2d to (1, 0) and Linear layers to N(0, 0.01); None / 'default' keeps the constructors' own initialisation
  (reference behaviour: initModel.py:9-32)."""
  if initType in (None, 'default'):
    return
  draw = _WEIGHT_INIT.get(initType)
  for module in model.modules():
    if isinstance(module, _CONV_TYPES):
      if draw is not None:
        draw(module.weight)
      _zero_bias(module)
    elif isinstance(module, (nn.BatchNorm1d, nn.BatchNorm2d)):
      nn.init.constant_(module.weight, 1)
      nn.init.constant_(module.bias, 0)
    elif isinstance(module, nn.Linear):
      nn.init.normal_(module.weight, 0, 0.01)
      _zero_bias(module)


def loadStackHourglassOnly(model, savedDictPath):
  """Partial warm start from a (PSMNet) checkpoint: every saved entry that `model` also has is taken over, except the feature
  extractor's ('feature_extraction' / 'forfilter1' in the key); everything else keeps its current value (initModel.py:35-42)."""
  checkpoint = torch.load(savedDictPath)['state_dict']
  merged = model.state_dict()
  skip = ('feature_extraction', 'forfilter1')
  taken = {key: value for key, value in checkpoint.items() if key in merged and not any(tag in key for tag in skip)}
  merged.update(taken)
  print("load partial parameter: ")
  model.load_state_dict(merged)
  print("loading done!")
