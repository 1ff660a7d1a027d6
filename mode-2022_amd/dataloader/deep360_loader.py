"""Deep360 datasets for the two stages (public surface of the reference's dataloader/deep360_loader.py:60-167: the class names,
constructor arguments and item formats the training / test scripts rely on).

Written against PIL + numpy (the reference needs cv2 and torchvision, neither present here), which also makes it **parity
unpinned**: the reference module cannot be imported in this environment, so tests/test_dataloader.py checks the documented
behaviour on a synthetic tree instead.
"""
import random

import numpy as np
import torch
from PIL import Image
from torch.utils.data import Dataset

from . import preprocess

CROP_H, CROP_W = 512, 256  # training crop of the disparity stage (deep360_loader.py:103)


# ---- file readers (same names as the reference's module-level loaders) -------------------------------------------------
def _npz_float32(path):
  with np.load(path) as z:
    return z['arr_0'].astype(np.float32)


def default_loader(path):
  """RGB panorama as a PIL image."""
  with Image.open(path) as im:
    return im.convert('RGB')


def disparity_loader(path):
  """(H, W) float32 disparity map stored as arr_0 of an .npz."""
  return _npz_float32(path)


def depth_loader(path):
  """(H, W, 1) float32 depth map."""
  return _npz_float32(path)[..., None]


def conf_loader(path):
  """(1, H, W) float32 confidence in [0, 1] from an 8-bit image.  The reference takes channel 0 of cv2.imread (blue); the
  exported confidence maps are grey, so any channel carries the same value -- the blue one is used here as well."""
  blue = np.asarray(default_loader(path))[:, :, 2]
  return (blue / 255.0).astype(np.float32)[None]


def resize_nearest(a, width, height):
  """What cv2.resize(a, (width, height), interpolation=cv2.INTER_NEAREST) returns: destination pixel (y, x) copies source
  pixel (floor(y * H / height), floor(x * W / width)), clamped to the image."""
  src_h, src_w = a.shape[:2]
  rows = np.minimum(np.floor(np.arange(height) * (src_h / height)).astype(np.int64), src_h - 1)
  cols = np.minimum(np.floor(np.arange(width) * (src_w / width)).astype(np.int64), src_w - 1)
  return a[np.ix_(rows, cols)] if a.ndim == 2 else a[rows][:, cols]


class Deep360DatasetDisparity(Dataset):
  """One stereo pair per item: {'leftImg': (3,H,W), 'rightImg': (3,H,W), 'dispMap': (1,H,W), 'dispNames': path}.

  Pairs stored at another width than `shape` are resized (images with PIL, the disparity map by nearest neighbour and multiplied
  by the width ratio, deep360_loader.py:96-99).  crop=True cuts the same random 512 x 256 window out of all three; the
  reference's own crop branch (:101-108) uses names it never defined and raises."""

  def __init__(self, leftImgs, rightImgs, disps, shape=(1024, 512), crop=False, disploader=disparity_loader, rgbloader=default_loader):
    super(Deep360DatasetDisparity, self).__init__()
    self.leftImgs, self.rightImgs, self.disps = leftImgs, rightImgs, disps
    self.height, self.width = shape
    self.crop = crop
    self.disp_loader, self.rgb_loader = disploader, rgbloader
    self.processed = preprocess.get_transform_stage1(augment=False)

  def __len__(self):
    return len(self.disps)

  def _fit(self, views, disp):
    """Bring a pair and its disparity map to the working size."""
    stored_w = views[0].size[0]
    if stored_w == self.width:
      return views, disp
    size = (self.width, self.height)
    return [v.resize(size) for v in views], resize_nearest(disp, *size) * (self.width / stored_w)

  @staticmethod
  def _window(views, disp):
    full_w, full_h = views[0].size
    left, top = random.randint(0, full_w - CROP_W), random.randint(0, full_h - CROP_H)
    box = (left, top, left + CROP_W, top + CROP_H)
    return [v.crop(box) for v in views], disp[top:top + CROP_H, left:left + CROP_W]

  def __getitem__(self, index):
    name = self.disps[index]
    views = [self.rgb_loader(self.leftImgs[index]), self.rgb_loader(self.rightImgs[index])]
    views, disp = self._fit(views, self.disp_loader(name))
    if self.crop:
      views, disp = self._window(views, disp)
    disp = torch.from_numpy(np.ascontiguousarray(disp, dtype=np.float32))[None]
    return {'leftImg': self.processed(views[0]), 'rightImg': self.processed(views[1]), 'dispMap': disp, 'dispNames': name}


class Deep360DatasetFusion(Dataset):
  """One frame per item, as the fusion scripts unpack it: (ground-truth path, [6 depth tensors (1,H,W)], [6 confidence arrays
  (1,H,W)], [4 rgb tensors (3,H,W)], ground truth (H,W)) (deep360_loader.py:120-167).  resize=True halves every input by
  taking every second pixel (the panoramas through PIL); the ground truth is halved only when training."""

  def __init__(self, depthes, confs, rgbs, gt, resize, training, depthloader=depth_loader, rgbloader=default_loader):
    super(Deep360DatasetFusion, self).__init__()
    self.depthes, self.confs, self.rgbs, self.gt = depthes, confs, rgbs, gt
    self.resize, self.training = resize, training
    self.depthloader, self.rgbloader = depthloader, rgbloader

  def __len__(self):
    return len(self.depthes[0])  # as the reference (deep360_loader.py: the first depth list, not the ground truth list)

  def __getitem__(self, index):
    depth_maps = [self.depthloader(paths[index]) for paths in self.depthes]
    conf_maps = [conf_loader(paths[index]) for paths in self.confs]
    panoramas = [self.rgbloader(paths[index]) for paths in self.rgbs]
    truth = np.ascontiguousarray(self.depthloader(self.gt[index])[..., 0], dtype=np.float32)
    if self.resize:
      depth_maps = [m[::2, ::2] for m in depth_maps]
      conf_maps = [m[:, ::2, ::2] for m in conf_maps]
      half = (panoramas[0].size[0] // 2, panoramas[0].size[1] // 2)
      panoramas = [im.resize(half) for im in panoramas]
      if self.training:
        truth = truth[::2, ::2]
    as_depth = preprocess.get_transform_stage2(augment=False)
    as_rgb = preprocess.get_transform_stage1(augment=False)
    return self.gt[index], [as_depth(m) for m in depth_maps], conf_maps, [as_rgb(im) for im in panoramas], truth
