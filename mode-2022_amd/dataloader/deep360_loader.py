"""Deep360 datasets (reference dataloader/deep360_loader.py:60-167): same class names, constructor arguments and item
formats.  PIL + numpy instead of cv2 / torchvision; **parity unpinned** (the reference file cannot be imported here: cv2 and
torchvision are absent), the tests check the documented behaviour on a synthetic tree."""
import random

import numpy as np
import torch
from PIL import Image
from torch.utils.data import Dataset

from . import preprocess


def default_loader(path):
  return Image.open(path).convert('RGB')


def disparity_loader(path):
  return np.load(path)['arr_0'].astype(np.float32)


def depth_loader(path):
  return np.expand_dims(np.load(path)['arr_0'].astype(np.float32), axis=-1)


def conf_loader(path):
  """(1, H, W) float32 in [0, 1] from the first stored channel of an 8-bit image (deep360_loader.py:27-29 reads it with
  cv2.imread, whose channel 0 is BLUE; the exported confidence maps are grey, all channels equal)."""
  img = np.asarray(Image.open(path).convert('RGB'))
  return np.expand_dims((img[:, :, 2] / 255.0).astype(np.float32), axis=0)


def resize_nearest(a, width, height):
  """cv2.resize(a, (width, height), interpolation=cv2.INTER_NEAREST): source index = floor(dst * src / dst_size), clamped."""
  h, w = a.shape[:2]
  ys = np.minimum((np.arange(height) * (h / height)).astype(np.int64), h - 1)
  xs = np.minimum((np.arange(width) * (w / width)).astype(np.int64), w - 1)
  return a[ys][:, xs]


class Deep360DatasetDisparity(Dataset):
  """Items: {'leftImg' (3,H,W), 'rightImg' (3,H,W), 'dispMap' (1,H,W), 'dispNames'} (deep360_loader.py:60-117).
  Images of another width are resized to `shape` (disparities by nearest neighbour, scaled by the width ratio, :96-99).
  crop=True takes a random 512x256 window; the reference's branch (:101-108) refers to undefined names and raises -- here it
  crops left, right and disparity consistently."""

  def __init__(self, leftImgs, rightImgs, disps, shape=(1024, 512), crop=False, disploader=disparity_loader, rgbloader=default_loader):
    super(Deep360DatasetDisparity, self).__init__()
    self.crop = crop
    self.height, self.width = shape
    self.processed = preprocess.get_transform_stage1(augment=False)
    self.leftImgs, self.rightImgs, self.disps = leftImgs, rightImgs, disps
    self.disp_loader, self.rgb_loader = disploader, rgbloader

  def __getitem__(self, index):
    disp_name = self.disps[index]
    left = self.rgb_loader(self.leftImgs[index])
    right = self.rgb_loader(self.rightImgs[index])
    disp = self.disp_loader(disp_name)
    w, h = left.size
    if w != self.width:
      left = left.resize((self.width, self.height))
      right = right.resize((self.width, self.height))
      disp = resize_nearest(disp, self.width, self.height) * (self.width / w)
    if self.crop:
      w, h = left.size
      th, tw = 512, 256
      x1, y1 = random.randint(0, w - tw), random.randint(0, h - th)
      left = left.crop((x1, y1, x1 + tw, y1 + th))
      right = right.crop((x1, y1, x1 + tw, y1 + th))
      disp = disp[y1:y1 + th, x1:x1 + tw]
    disp = np.ascontiguousarray(disp, dtype=np.float32)
    return {'leftImg': self.processed(left), 'rightImg': self.processed(right), 'dispMap': torch.from_numpy(disp).unsqueeze_(0),
            'dispNames': disp_name}

  def __len__(self):
    return len(self.disps)


class Deep360DatasetFusion(Dataset):
  """Items: (gt name, 6 depth tensors (1,H,W), 6 confidence arrays (1,H,W), 4 rgb tensors (3,H,W), gt (H,W))
  (deep360_loader.py:120-167); resize=True halves everything by striding (rgb by PIL resize), the ground truth only in
  training."""

  def __init__(self, depthes, confs, rgbs, gt, resize, training, depthloader=depth_loader, rgbloader=default_loader):
    super(Deep360DatasetFusion, self).__init__()
    self.depthes, self.confs, self.rgbs, self.gt = depthes, confs, rgbs, gt
    self.depthloader, self.rgbloader = depthloader, rgbloader
    self.resize, self.training = resize, training

  def __getitem__(self, index):
    depthes = [self.depthloader(d[index]) for d in self.depthes]
    confs = [conf_loader(c[index]) for c in self.confs]
    rgbs = [self.rgbloader(r[index]) for r in self.rgbs]
    gt = np.ascontiguousarray(np.squeeze(self.depthloader(self.gt[index]), axis=-1), dtype=np.float32)
    if self.resize:
      depthes = [d[::2, ::2, :] for d in depthes]
      confs = [c[:, ::2, ::2] for c in confs]
      w, h = rgbs[0].size
      rgbs = [r.resize((int(w / 2), int(h / 2))) for r in rgbs]
      if self.training:
        gt = gt[::2, ::2]
    to_depth = preprocess.get_transform_stage2(augment=False)
    to_rgb = preprocess.get_transform_stage1(augment=False)
    return self.gt[index], [to_depth(d) for d in depthes], confs, [to_rgb(r) for r in rgbs], gt

  def __len__(self):
    return len(self.gt)
