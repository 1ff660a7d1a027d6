"""Deep360 data access: file lists against the reference's own list_file.py (golden JSON made by
tests/golden/make_golden_lists.py on the miniature tree of deep360_tree.py), datasets by construction (the reference loader
needs cv2 + torchvision and cannot be imported: parity unpinned)."""
import json
import os

import numpy as np
import pytest
import torch
from PIL import Image

import dataloader
import deep360_tree
from conftest import GOLDEN
from dataloader import deep360_loader, list_file, preprocess


def _rel(obj, root):
  return os.path.relpath(obj, root) if isinstance(obj, str) else [_rel(o, root) for o in obj]


@pytest.mark.parametrize('soiled', [False, True])
def test_file_lists_match_the_reference(tmp_path, soiled):
  gold = json.load(open(os.path.join(GOLDEN, 'deep360_lists.json')))
  dataset, exported, _ = deep360_tree.build(str(tmp_path))
  tag = 'soiled' if soiled else 'clean'
  root = str(tmp_path)
  assert _rel(list(list_file.list_deep360_disparity_train(dataset, soiled)), root) == gold['disparity_train/' + tag]
  assert _rel(list(list_file.list_deep360_disparity_test(dataset, soiled)), root) == gold['disparity_test/' + tag]
  assert _rel(list(list_file.list_deep360_fusion_train(exported, dataset, soiled)), root) == gold['fusion_train/' + tag]
  assert _rel(list(list_file.list_deep360_fusion_test(exported, dataset, soiled)), root) == gold['fusion_test/' + tag]
  # the pairing the stage relies on: left/right/disparity of the same frame and camera pair
  left, right, disp = list_file.list_deep360_disparity_test(dataset, soiled)
  assert len(left) == len(right) == len(disp) == 6 * 2 * 6
  for l, r, d in zip(left, right, disp):
    key = os.path.basename(d)[:len('ep1_000007_12')]
    assert os.path.basename(l).startswith(key) and os.path.basename(r).startswith(key) and l != r


def test_missing_directory_raises_like_the_reference(tmp_path):
  with pytest.raises(FileNotFoundError):  # os.listdir of a missing directory, list_file.py:51
    list_file.list_deep360_disparity_test(str(tmp_path), False)


def test_transforms():
  rng = np.random.RandomState(0)
  img = rng.randint(0, 256, (6, 4, 3)).astype(np.uint8)
  t = preprocess.get_transform_stage1(augment=False)(Image.fromarray(img))
  assert t.dtype == torch.float32 and tuple(t.shape) == (3, 6, 4)
  ref = (img.astype(np.float32).transpose(2, 0, 1) / 255 - np.array([0.485, 0.456, 0.406], np.float32)[:, None, None]) / \
      np.array([0.229, 0.224, 0.225], np.float32)[:, None, None]
  assert np.abs(t.numpy() - ref).max() < 1e-6
  depth = rng.rand(6, 4, 1).astype(np.float32) * 100
  d = preprocess.get_transform_stage2()(depth)
  assert tuple(d.shape) == (1, 6, 4) and np.array_equal(d.numpy()[0], depth[:, :, 0])  # float input is not rescaled
  with pytest.raises(NotImplementedError):
    preprocess.get_transform_stage1(augment=True)


def test_resize_nearest_is_the_floor_rule():
  a = np.arange(6 * 8, dtype=np.float32).reshape(6, 8)
  assert np.array_equal(deep360_loader.resize_nearest(a, 4, 3), a[::2, ::2])          # exact halving
  up = deep360_loader.resize_nearest(a, 16, 12)
  assert np.array_equal(up, np.repeat(np.repeat(a, 2, 0), 2, 1))                       # exact doubling
  odd = deep360_loader.resize_nearest(a, 5, 6)
  assert np.array_equal(odd[0], a[0, [0, 1, 3, 4, 6]])                                 # floor(x * 8 / 5)


def _write_pair(d, name, h, w, seed):
  rng = np.random.RandomState(seed)
  for side in ('a', 'b'):
    Image.fromarray(rng.randint(0, 256, (h, w, 3)).astype(np.uint8)).save(os.path.join(d, '%s_%s.png' % (name, side)))
  disp = rng.rand(h, w).astype(np.float64) * 40
  np.savez(os.path.join(d, name + '_disp.npz'), disp)
  return disp


def test_disparity_dataset_items(tmp_path):
  d = str(tmp_path)
  disp_full = _write_pair(d, 'f0', 32, 16, 1)
  ds = dataloader.Deep360DatasetDisparity([os.path.join(d, 'f0_a.png')], [os.path.join(d, 'f0_b.png')], [os.path.join(d, 'f0_disp.npz')],
                                          shape=(32, 16))
  assert len(ds) == 1
  it = ds[0]
  assert set(it) == {'leftImg', 'rightImg', 'dispMap', 'dispNames'} and it['dispNames'].endswith('f0_disp.npz')
  assert tuple(it['leftImg'].shape) == (3, 32, 16) and tuple(it['dispMap'].shape) == (1, 32, 16) and it['dispMap'].dtype == torch.float32
  assert np.array_equal(it['dispMap'].numpy()[0], disp_full.astype(np.float32))
  left = np.asarray(Image.open(os.path.join(d, 'f0_a.png')))
  assert abs(float(it['leftImg'][0, 3, 5]) - (left[3, 5, 0] / 255 - 0.485) / 0.229) < 1e-6
  # another working size: images resized, disparities by nearest neighbour and scaled by the width ratio
  half = dataloader.Deep360DatasetDisparity(ds.leftImgs, ds.rightImgs, ds.disps, shape=(16, 8))[0]
  assert tuple(half['leftImg'].shape) == (3, 16, 8)
  assert np.allclose(half['dispMap'].numpy()[0], disp_full.astype(np.float32)[::2, ::2] * 0.5)
  # batches through the stock DataLoader, as train_disparity.py builds them
  batch = next(iter(torch.utils.data.DataLoader(ds, batch_size=1)))
  assert tuple(batch['leftImg'].shape) == (1, 3, 32, 16) and tuple(batch['dispMap'].shape) == (1, 1, 32, 16)


def test_disparity_dataset_crop(tmp_path):
  d = str(tmp_path)
  disp_full = _write_pair(d, 'f1', 600, 300, 2)
  ds = dataloader.Deep360DatasetDisparity([os.path.join(d, 'f1_a.png')], [os.path.join(d, 'f1_b.png')], [os.path.join(d, 'f1_disp.npz')],
                                          shape=(600, 300), crop=True)
  it = ds[0]
  assert tuple(it['leftImg'].shape) == (3, 512, 256) and tuple(it['dispMap'].shape) == (1, 512, 256)
  # the window is the same for the three arrays: find it from the disparity values
  got = it['dispMap'].numpy()[0]
  hits = np.argwhere(disp_full.astype(np.float32) == got[0, 0])
  y1, x1 = hits[0]
  assert np.array_equal(got, disp_full.astype(np.float32)[y1:y1 + 512, x1:x1 + 256])
  left = np.asarray(Image.open(os.path.join(d, 'f1_a.png')))
  assert abs(float(it['leftImg'][1, 0, 0]) - (left[y1, x1, 1] / 255 - 0.456) / 0.224) < 1e-6


def test_fusion_dataset_items(tmp_path):
  d = str(tmp_path)
  rng = np.random.RandomState(3)
  depthes, confs, rgbs = [], [], []
  for p in range(6):
    np.savez(os.path.join(d, 'depth%d.npz' % p), rng.rand(8, 4) * 50)
    depthes.append([os.path.join(d, 'depth%d.npz' % p)])
    grey = rng.randint(0, 256, (8, 4)).astype(np.uint8)
    Image.fromarray(np.stack([grey] * 3, -1)).save(os.path.join(d, 'conf%d.png' % p))
    confs.append([os.path.join(d, 'conf%d.png' % p)])
  for k in range(4):
    Image.fromarray(rng.randint(0, 256, (8, 4, 3)).astype(np.uint8)).save(os.path.join(d, 'rgb%d.png' % k))
    rgbs.append([os.path.join(d, 'rgb%d.png' % k)])
  gt = rng.rand(8, 4) * 50
  np.savez(os.path.join(d, 'gt.npz'), gt)
  full = dataloader.Deep360DatasetFusion(depthes, confs, rgbs, [os.path.join(d, 'gt.npz')], resize=False, training=True)
  name, dd, cc, rr, g = full[0]
  assert name.endswith('gt.npz') and len(dd) == 6 and len(cc) == 6 and len(rr) == 4 and len(full) == 1
  assert tuple(dd[0].shape) == (1, 8, 4) and tuple(cc[0].shape) == (1, 8, 4) and tuple(rr[0].shape) == (3, 8, 4) and g.shape == (8, 4)
  assert np.array_equal(dd[2].numpy()[0], np.load(depthes[2][0])['arr_0'].astype(np.float32))
  grey = np.asarray(Image.open(confs[1][0]).convert('RGB'))[:, :, 0]
  assert np.allclose(cc[1][0], grey / 255.0) and cc[1].dtype == np.float32
  assert np.array_equal(g, gt.astype(np.float32))
  for training in (True, False):
    _, dd2, cc2, rr2, g2 = dataloader.Deep360DatasetFusion(depthes, confs, rgbs, [os.path.join(d, 'gt.npz')], resize=True, training=training)[0]
    assert tuple(dd2[0].shape) == (1, 4, 2) and tuple(cc2[0].shape) == (1, 4, 2) and tuple(rr2[0].shape) == (3, 4, 2)
    assert np.array_equal(dd2[0].numpy()[0], dd[0].numpy()[0][::2, ::2])
    assert g2.shape == ((4, 2) if training else (8, 4))  # the ground truth is halved in training only (deep360_loader.py:154-155)
