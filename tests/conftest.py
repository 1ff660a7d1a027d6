import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
PKG = os.path.join(ROOT, 'mode-2022_amd')
for p in (ROOT, GOLDEN, PKG):
  if p not in sys.path:
    sys.path.insert(0, p)


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
  config.addinivalue_line('markers', 'f16_pricing: tests/test_bench_math.py -- keeps the two-piece fp16 pricing of the stride-1 3-D labels')


def pytest_collection_modifyitems(config, items):
  # GPU tests are never silently skipped on a GPU box; on a CPU-only host they are deselected by
  # `-m "not gpu"`, and if someone runs them anyway they fail loudly in the native loader.
  pass


@pytest.fixture(scope='session', autouse=True)
def _native_library():
  """The .so is a build artefact (git-ignored): build it in-tree if it is missing or stale (hipcc cross-compiles
  gfx950 without a GPU; a no-op when up to date).  Tests never run against anything but this library."""
  from mode_hip import build as hip_build
  hip_build.build(force=False, verbose=False)


@pytest.fixture(scope='session')
def golden():
  def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
  return load


@pytest.fixture(autouse=True)
def _threads():
  torch.set_num_threads(min(8, os.cpu_count() or 1))


@pytest.fixture(params=['f32', 'bf16x6'])
def arith(request):
  """Both arithmetics of the stride-1 3x3x3 layers (mode_hip.functional.CONV_ARITH): fp32 MFMA and the split-bf16 matrix path
  (the default).  Tests of those layers take this fixture, so neither kernel family goes untested whichever is the default."""
  from mode_hip import functional as HF
  HF.set_conv_arith(request.param)
  yield request.param
  HF.set_conv_arith('bf16x6')
