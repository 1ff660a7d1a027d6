"""Input normalisation (reference dataloader/preprocess.py:8, 49-76): ToTensor + Normalize, without torchvision.

Only the non-augmented transforms are on the path (Deep360DatasetDisparity / Deep360DatasetFusion build them with
augment=False, deep360_loader.py:77, 157, 161); the colour-jitter + PCA-lighting augmentation of preprocess.py:33-46 is not
provided and asking for it raises."""
import numpy as np
import torch

imagenet_stats = {'mean': [0.485, 0.456, 0.406], 'std': [0.229, 0.224, 0.225]}
deep360_stats = {'mean': [0], 'std': [1]}


def to_tensor(pic):
  """torchvision's ToTensor: PIL image or (H, W, C) array -> float32 (C, H, W); uint8 input is scaled by 1/255, other dtypes
  are kept as they are (so float32 depth maps pass through unscaled, as in the reference)."""
  a = np.asarray(pic)
  if a.ndim == 2:
    a = a[:, :, None]
  t = torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))
  return t.to(torch.float32).div(255) if a.dtype == np.uint8 else t


class _Normalize(object):
  def __init__(self, mean, std):
    self.mean, self.std = mean, std

  def __call__(self, pic):
    t = to_tensor(pic)
    mean = torch.as_tensor(self.mean, dtype=t.dtype).view(-1, 1, 1)
    std = torch.as_tensor(self.std, dtype=t.dtype).view(-1, 1, 1)
    return (t - mean) / std


def color_normalize(normalize=imagenet_stats):
  return _Normalize(**normalize)


def depth_normalize(normalize=deep360_stats):
  return _Normalize(**normalize)


def get_transform_stage1(name='imagenet', normalize=None, augment=True):
  """RGB transform (preprocess.py:64-69); `normalize` is ignored there too (always the ImageNet statistics)."""
  if augment:
    raise NotImplementedError('the colour augmentation of preprocess.py:33-46 is not part of this build')
  return color_normalize(imagenet_stats)


def get_transform_stage2(name='deep360', normalize=None, augment=False):
  """Depth transform (preprocess.py:72-74): to tensor, mean 0 / std 1."""
  return depth_normalize(deep360_stats)
