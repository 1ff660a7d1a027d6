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


def initModelPara(model, initType):
  """initModel.py:9-32: no-op for None / 'default'; otherwise re-draws every conv-like weight."""
  if initType is None or initType == 'default':
    return
  for m in model.modules():
    if isinstance(m, _CONV_TYPES):
      if initType in _WEIGHT_INIT:
        _WEIGHT_INIT[initType](m.weight)
      if m.bias is not None:
        nn.init.constant_(m.bias, 0)
    elif isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)):
      nn.init.constant_(m.weight, 1)
      nn.init.constant_(m.bias, 0)
    elif isinstance(m, nn.Linear):
      nn.init.normal_(m.weight, 0, 0.01)
      if m.bias is not None:
        nn.init.constant_(m.bias, 0)


def loadStackHourglassOnly(model, savedDictPath):
  """initModel.py:35-42: take every entry of a (PSMNet) checkpoint that exists in `model` and does not belong
  to the feature extractor; keep the rest of the current parameters."""
  saved = torch.load(savedDictPath)['state_dict']
  current = model.state_dict()
  current.update({k: v for k, v in saved.items() if k in current and 'feature_extraction' not in k and 'forfilter1' not in k})
  print("load partial parameter: ")
  model.load_state_dict(current)
  print("loading done!")
