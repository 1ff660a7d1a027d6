#!/usr/bin/env python3
"""Golden vectors for the export-stage geometry, made by the REFERENCE itself in the dev container.

  python tests/golden/make_golden_geometry.py      # writes tests/golden/geometry.npz

Harness stand-ins (no reference file is edited or copied): a `numba` module whose `jit` decorator returns the function
unchanged (numba is not installed; the decorated loop is plain Python), and an identity `torch.Tensor.cuda`
(utils/geometry.py hard-codes `.cuda()`; there is no GPU here).  `disp2depth` lives in save_output_disparity_stage.py, a
script that cannot be imported here (argparse at import time, torchvision / cv2 / dataset loaders absent): its own arithmetic
(the sine rule, 15 lines) is NOT pinned by the reference -- only the functions it calls are."""
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'


def main():
  numba = types.ModuleType('numba')
  numba.jit = lambda *a, **k: (lambda f: f)
  sys.modules['numba'] = numba
  torch.Tensor.cuda = lambda self, *a, **k: self
  sys.path.insert(0, REF)
  geo = importlib.import_module('utils.geometry')

  rng = np.random.RandomState(20221)
  out = {}
  # depth maps with holes (zeros), a wide dynamic range and exact duplicates (ties in the z-buffer)
  for tag, (h, w) in (('a', (64, 32)), ('b', (128, 64)), ('c', (50, 24))):
    depth = (rng.rand(h, w).astype(np.float32) * 8 + 0.3)
    depth[rng.rand(h, w) < 0.1] = 0
    depth[rng.rand(h, w) < 0.05] = 1000
    depth = np.round(depth * 4) / 4 if tag == 'c' else depth  # quantised: many equal radii
    conf = rng.rand(h, w).astype(np.float32)
    out[tag + '/depth'], out[tag + '/conf'] = depth, conf
    for name, args in (('t23', (0, -np.sqrt(2) / 2, -np.sqrt(2) / 2, 0.75 * np.pi, 0, 0)), ('t24', (0, -1, 0, 0.5 * np.pi, 0, 0)),
                       ('t34', (0, 1, 0, 0, 0, 0)), ('tid', (0, 0, 0, 0, 0, 0)), ('tgen', (0.3, -0.2, 0.5, 0.4, -0.7, 0.2))):
      v2, c2 = geo.depthViewTransWithConf(depth.copy(), conf.copy(), *args)
      out['%s/%s/view' % (tag, name)], out['%s/%s/conf' % (tag, name)] = v2, c2
      out['%s/%s/args' % (tag, name)] = np.array(args, dtype=np.float64)
    img = rng.rand(h, w, 3).astype(np.float32)
    out[tag + '/img'] = img
    out[tag + '/rot13'] = geo.rotateCassini(img, 0.5 * np.pi, 0, 0)
    out[tag + '/rot_gen'] = geo.rotateCassini(img, 0.3, -0.4, 1.1)
    if h == 2 * w:
      out[tag + '/c2e'] = geo.cassini2Equirec(img)
      erp = rng.rand(w, h, 3).astype(np.float32)
      R = geo.np.array([[np.cos(0.4), -np.sin(0.4), 0], [np.sin(0.4), np.cos(0.4), 0], [0, 0, 1.0]])
      out[tag + '/erp'], out[tag + '/e2c_R'] = erp, R
      out[tag + '/e2c'] = geo.erp2rect_cassini(erp, R, h, w, 'cpu')

  np.savez_compressed(os.path.join(HERE, 'geometry.npz'), **out)
  print('wrote geometry.npz with %d arrays' % len(out))


if __name__ == '__main__':
  main()
