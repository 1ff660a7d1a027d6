"""Deterministic weight / input recipes shared by the fixture generator and the tests.

Weights are never committed (22 MB); both sides regenerate them from this recipe: walk the
(key, shape) manifest in state_dict order and fill every tensor from one
``numpy.random.RandomState(seed)`` stream with a per-kind scale.
"""
import json
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
IMAGENET_MEAN = np.array([0.485, 0.456, 0.406], np.float32).reshape(1, 3, 1, 1)
IMAGENET_STD = np.array([0.229, 0.224, 0.225], np.float32).reshape(1, 3, 1, 1)


def load_manifest(name='manifest_mode_disparity.json'):
  with open(os.path.join(HERE, name)) as f:
    return [(k, tuple(s)) for k, s in json.load(f)]


def recipe_tensor(rs, key, shape):
  if key.endswith('num_batches_tracked'):
    return np.zeros(shape, np.int64)
  if key.endswith('running_mean'):
    return np.zeros(shape, np.float32)
  if key.endswith('running_var'):
    return np.ones(shape, np.float32)
  if len(shape) >= 4:  # conv / sphere-conv / transposed-conv weight
    fan = float(np.prod(shape[2:])) * shape[0]
    return (rs.standard_normal(shape) * np.sqrt(2.0 / fan)).astype(np.float32)
  if key.endswith('.weight'):  # BN gamma
    return (1.0 + 0.1 * rs.standard_normal(shape)).astype(np.float32)
  if key.endswith('.bias'):  # BN beta
    return (0.1 * rs.standard_normal(shape)).astype(np.float32)
  raise KeyError(key)


def recipe_state(manifest, seed, dtype=torch.float32):
  rs = np.random.RandomState(seed)
  out = {}
  for key, shape in manifest:
    t = torch.from_numpy(recipe_tensor(rs, key, shape))
    out[key] = t.to(dtype) if t.is_floating_point() else t
  return out


def recipe_images(B, H, W, seed, shift=3):
  """Left = ImageNet-normalised uniform noise; right = left rolled by `shift` px along W + noise, so
  that the cost volume has a matching structure.  Returns float32 tensors (B,3,H,W)."""
  rs = np.random.RandomState(seed)
  left = (rs.rand(B, 3, H, W).astype(np.float32) - IMAGENET_MEAN) / IMAGENET_STD
  right = np.roll(left, -shift, axis=3) + 0.01 * rs.standard_normal((B, 3, H, W)).astype(np.float32)
  return torch.from_numpy(left), torch.from_numpy(right.astype(np.float32))


def recipe_disparity(B, H, W, seed, maxdisp):
  """Synthetic ground truth with 5 % NaN (exercises the mask of train_disparity.py:195)."""
  rs = np.random.RandomState(seed)
  d = (rs.rand(B, 1, H, W).astype(np.float32)) * (maxdisp / 2.0)
  d[rs.rand(B, 1, H, W) < 0.05] = np.nan
  return torch.from_numpy(d)
