#!/usr/bin/env python3
"""tests/golden/model_wc_config4_eval.npz: the INFERENCE forward (and the training-mode forward) of one 2048 x 1024 pair at 256 disparities (BASELINE configs[4], the
per-GPU share of its batch) evaluated by the CPU oracle in float64.

Unlike model_wc_{tiny,cfg1,full}.npz this fixture is NOT made by the imported reference: at this size its CPU run needs more memory
than the development container has (64 GiB), and a float64 evaluation is what a parity bound of 1e-3 px should be held against anyway.
It is a stored result of oracle/mode_ref.py -- the restatement that the reference's own fixtures pin at 64 x 32, 512 x 256 and
1024 x 512 (tests/test_oracle_golden.py) -- so that the GPU tier does not spend nine CPU minutes per run recomputing it
(tests/test_gpu_parity.py::test_eval_output_at_config4_size_against_the_float64_oracle).  Nothing in here touches a GPU; it was run on
the host CPUs of the MI355X box (16 cores, ~9 min, ~40 GB):

    gpurun -- 'python tests/golden/make_golden_config4_eval.py gpurun_out/model_wc_config4_eval.npz'

Same layout as the model_wc_* fixtures (recipe.fixture_state / fixture_inputs read it):
  cfg = [maxdisp, H, W, B, seed], wc = [mix, logit_scale], shift, sub
  bn/<key>            running statistics := the batch statistics of the same pair (one train-mode forward of the oracle with momentum 1),
                      rounded to float32 -- the values the eval pass below uses and the test loads
  train/pred{1,2,3}(_block)  the three predictions of the TRAINING-mode forward (batch statistics; the calibration pass itself), as below
  eval/pred3          the eval-mode output (train_disparity.py:167-170: the last prediction), every `sub`-th pixel, float64
  eval/pred3_block    8 x 8 block means of ALL pixels (float64)
"""
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import recipe  # noqa: E402
from oracle import mode_ref  # noqa: E402

MAXDISP, H, W, B, SEED, SUB, SHIFT, LOGIT_SCALE = 256, 2048, 1024, 1, 700, 8, 3, 0.015


def usable_cores():
  n = len(os.sched_getaffinity(0))
  try:
    with open('/sys/fs/cgroup/cpu.max') as f:
      quota, period = f.read().split()
    if quota != 'max':
      n = min(n, max(1, int(float(quota) / float(period))))
  except (OSError, ValueError):
    pass
  return n


def main():
  out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, 'model_wc_config4_eval.npz')
  torch.set_num_threads(usable_cores())
  t0 = time.time()
  state = recipe.recipe_state_wc(recipe.load_manifest(), SEED, logit_scale=LOGIT_SCALE)
  left, right = recipe.recipe_images(B, H, W, SEED + 1, shift=SHIFT)
  P = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in state.items()}
  pos = mode_ref.sphere_position(H // 4, W // 4, 'Cassini')
  out = dict(cfg=np.array([MAXDISP, H, W, B, SEED]), sub=np.array(SUB), wc=np.array([recipe.WC_MIX, LOGIT_SCALE]), shift=np.array(SHIFT))
  keep = mode_ref.BN_MOMENTUM
  mode_ref.BN_MOMENTUM = 1.0  # the calibration pass of make_golden_wc.py: running statistics := this batch's
  try:
    with torch.no_grad():
      tr = mode_ref.mode_disparity(P, left.double(), right.double(), MAXDISP, pos, True)
  finally:
    mode_ref.BN_MOMENTUM = keep
  for i, p in enumerate(tr):  # the calibration pass IS a training-mode forward: its three predictions come for free
    out['train/pred%d' % (i + 1)] = p[:, :, ::SUB, ::SUB].numpy().astype(np.float32)  # (float32: 8e-6 px at 128 px, and half the file)
    out['train/pred%d_block' % (i + 1)] = F.avg_pool2d(p.double(), 8).numpy().astype(np.float32)
  del tr
  print('calibration pass done (%.0f s)' % (time.time() - t0), flush=True)
  for k in list(P):
    if k.endswith('running_mean') or k.endswith('running_var'):
      out['bn/' + k] = P[k].float().numpy().copy()
      P[k] = torch.from_numpy(out['bn/' + k]).double()  # the eval pass uses exactly what the test will load
  with torch.no_grad():
    pred = mode_ref.mode_disparity(P, left.double(), right.double(), MAXDISP, pos, False)
  pred = pred[-1] if isinstance(pred, (list, tuple)) else pred
  out['eval/pred3'] = pred[:, :, ::SUB, ::SUB].numpy().copy()
  out['eval/pred3_block'] = F.avg_pool2d(pred.double(), 8).numpy()
  np.savez_compressed(out_path, **out)
  print('eval pass done (%.0f s): prediction %.2f .. %.2f px, mean %.2f; wrote %s (%.0f KB)' %
        (time.time() - t0, float(pred.min()), float(pred.max()), float(pred.mean()), out_path, os.path.getsize(out_path) / 1024.0), flush=True)


if __name__ == '__main__':
  main()
