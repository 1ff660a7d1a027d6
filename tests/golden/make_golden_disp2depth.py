#!/usr/bin/env python3
"""Golden vectors for disp2depth (save_output_disparity_stage.py:105-160), made by the REFERENCE's own function.

The script that holds it cannot be imported (argparse at import time; cv2 / torchvision / dataset loaders absent), so the function
is taken out of it with `ast` -- the FunctionDef node of `disp2depth`, compiled as it stands -- and run in a namespace that holds
what it refers to: numpy, math, the script-global `args.dbname`, and `rotateCassini` / `depthViewTransWithConf` of the imported
reference utils/geometry.py (same two stand-ins as make_golden_geometry.py: a pass-through `numba.jit`, an identity `.cuda()`).
No reference text is stored: the fixture holds inputs and outputs only.

  python tests/golden/make_golden_disp2depth.py      # writes tests/golden/disp2depth.npz   (development container only)
"""
import ast
import importlib
import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'


def reference_disp2depth():
  numba = types.ModuleType('numba')
  numba.jit = lambda *a, **k: (lambda f: f)
  sys.modules['numba'] = numba
  torch.Tensor.cuda = lambda self, *a, **k: self
  sys.path.insert(0, REF)
  geo = importlib.import_module('utils.geometry')
  path = os.path.join(REF, 'save_output_disparity_stage.py')
  tree = ast.parse(open(path).read(), path)
  fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == 'disp2depth']
  assert len(fn) == 1
  ns = dict(np=np, math=math, args=types.SimpleNamespace(dbname='Deep360'), rotateCassini=geo.rotateCassini,
            depthViewTransWithConf=geo.depthViewTransWithConf)
  exec(compile(ast.Module(body=fn, type_ignores=[]), path, 'exec'), ns)
  return ns['disp2depth'], ns['args']


def main():
  f, args = reference_disp2depth()
  rng = np.random.RandomState(7)
  H, W = 64, 32
  disp = (rng.rand(H, W).astype(np.float32) * 20)
  disp[rng.rand(H, W) < 0.1] = 0          # masked: depth 1000
  disp[rng.rand(H, W) < 0.05] = 1e-4      # tiny disparities: depth beyond 1000, clipped
  conf = rng.rand(H, W).astype(np.float32)
  out = dict(disp=disp, conf=conf)
  for dbname in ('Deep360', 'other'):
    args.dbname = dbname
    for pair in ('12', '13', '14', '23', '24', '34'):
      d, c = f(disp.copy(), conf.copy(), pair)
      out['%s/%s/depth' % (dbname, pair)] = np.asarray(d)
      out['%s/%s/conf' % (dbname, pair)] = np.asarray(c)
  np.savez_compressed(os.path.join(HERE, 'disp2depth.npz'), **out)
  print('wrote disp2depth.npz:', {k: (v.dtype, v.shape) for k, v in out.items() if k.endswith('12/depth')})


if __name__ == '__main__':
  main()
