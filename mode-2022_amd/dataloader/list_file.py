"""File lists of the Deep360 dataset (reference dataloader/list_file.py:30-267, same function names, arguments and
return tuples).

Layout (list_file.py:3-27): <root>/ep{1..6}_500frames/{training,validation,testing}/{rgb,rgb_soiled,disp,depth}; every frame
has 12 panoramas (pairs 12,13,14,23,24,34 as left/right), 6 disparity maps and 1 depth map; the fusion stage additionally reads
the exported <input>/ep*/<subset>/{disp_pred2depth,conf_map}[_soiled] (6 per frame).  All pairing is positional on the sorted
directory listings, exactly as in the reference.
"""
import os

EPISODES = ['ep%d_500frames' % i for i in range(1, 7)]


def _sorted_paths(directory):
  return [os.path.join(directory, n) for n in sorted(os.listdir(directory))]


def _disparity_subset(filepath, ep, subset, soiled):
  """(left images, right images, disparity maps) of one episode subset: disparity map i belongs to panoramas 2i, 2i+1
  (list_file.py:56-59)."""
  rgb = _sorted_paths(os.path.join(filepath, ep, subset, 'rgb_soiled' if soiled else 'rgb'))
  disp = _sorted_paths(os.path.join(filepath, ep, subset, 'disp'))
  n = len(disp)
  return [rgb[2 * i] for i in range(n)], [rgb[2 * i + 1] for i in range(n)], disp


def list_deep360_disparity_train(filepath, soiled):
  """-> train_left, train_right, train_disp, val_left, val_right, val_disp (list_file.py:30-65)."""
  out = {'training': ([], [], []), 'validation': ([], [], [])}
  for ep in sorted(EPISODES):
    for subset in ('training', 'validation'):
      for acc, part in zip(out[subset], _disparity_subset(filepath, ep, subset, soiled)):
        acc.extend(part)
  return out['training'] + out['validation']


def list_deep360_disparity_test(filepath, soiled):
  """-> test_left, test_right, test_disp (list_file.py:68-94)."""
  out = ([], [], [])
  for ep in sorted(EPISODES):
    for acc, part in zip(out, _disparity_subset(filepath, ep, 'testing', soiled)):
      acc.extend(part)
  return out


# panoramas of a frame used by the fusion stage: the left image of pair 12 (camera 1), its right image (camera 2) and the two
# images of pair 34 (cameras 3 and 4): positions 0, 1, 10, 11 of the frame's 12 files (list_file.py:171-174)
_FUSION_RGB = (0, 1, 10, 11)


def _fusion_subset(input_path, dataset_path, ep, subset, soil):
  """(6 depth lists, 6 confidence lists, 4 rgb lists, ground truth) of one episode subset (list_file.py:139-176)."""
  sfx = '_soiled' if soil else ''
  depth_in = _sorted_paths(os.path.join(input_path, ep, subset, 'disp_pred2depth' + sfx))
  conf_in = _sorted_paths(os.path.join(input_path, ep, subset, 'conf_map' + sfx))
  rgb = _sorted_paths(os.path.join(dataset_path, ep, subset, 'rgb' + sfx))
  gt = _sorted_paths(os.path.join(dataset_path, ep, subset, 'depth'))
  frames = range(len(gt))
  depthes = [[depth_in[6 * f + p] for f in frames] for p in range(6)]
  confs = [[conf_in[6 * f + p] for f in frames] for p in range(6)]
  rgbs = [[rgb[12 * f + k] for f in frames] for k in _FUSION_RGB]
  return depthes, confs, rgbs, gt


def _fusion_lists(input_path, dataset_path, soil, subsets):
  acc = {s: ([[] for _ in range(6)], [[] for _ in range(6)], [[] for _ in range(4)], []) for s in subsets}
  for ep in EPISODES:
    for s in subsets:
      depthes, confs, rgbs, gt = _fusion_subset(input_path, dataset_path, ep, s, soil)
      for dst, src in zip(acc[s][0] + acc[s][1] + acc[s][2], depthes + confs + rgbs):
        dst.extend(src)
      acc[s][3].extend(gt)
  return acc


def list_deep360_fusion_train(input_path, dataset_path, soil):
  """-> train_depthes, train_confs, train_rgbs, train_gt, val_depthes, val_confs, val_rgbs, val_gt (list_file.py:97-201)."""
  acc = _fusion_lists(input_path, dataset_path, soil, ('training', 'validation'))
  return tuple(acc['training']) + tuple(acc['validation'])


def list_deep360_fusion_test(input_path, dataset_path, soil):
  """-> test_depthes, test_confs, test_rgbs, test_gt (list_file.py:204-267)."""
  return tuple(_fusion_lists(input_path, dataset_path, soil, ('testing',))['testing'])
