"""CPU: the C-ABI library loads and exports exactly what include/mode_hip.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

import mode_hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
  src = open(os.path.join(ROOT, 'include', 'mode_hip.h')).read()
  src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
  return sorted(set(re.findall(r'\b(mode_[a-z0-9_]+)\s*\(', src)))


def test_header_declares_something():
  names = _declared()
  assert 'mode_cost_volume_fwd' in names and 'mode_sphere_conv_fwd' in names and len(names) >= 9


def test_every_declared_symbol_is_exported_and_bound():
  lib = mode_hip.lib()
  for name in _declared():
    assert hasattr(lib, name), 'libmode_hip.so does not export %s' % name
    assert name in mode_hip.SIGNATURES, 'ctypes binding missing for %s' % name
  assert sorted(mode_hip.SIGNATURES) == _declared()


def test_abi_version():
  assert mode_hip.lib().mode_hip_abi_version() == 1


def test_argument_validation_without_gpu():
  """Bad arguments are rejected on the host before any launch, with a message (reference: TORCH_CHECK)."""
  lib = mode_hip.lib()
  null = ctypes.c_void_p(0)
  rc = lib.mode_cost_volume_fwd(null, null, null, 1, 32, 48, 256, 128, null)
  assert rc == -1 and b'null pointer' in lib.mode_last_error()
  one = ctypes.c_void_p(16)
  rc = lib.mode_cost_volume_fwd(one, one, one, 1, 0, 48, 256, 128, null)
  assert rc == -1 and b'bad sizes' in lib.mode_last_error()
  rc = lib.mode_sphere_conv_fwd(one, one, one, one, one, 1, 6, 16, 32, 4, 3, 3, 1, 1, 16, 32, 4, null)
  assert rc == -1 and b'divisible' in lib.mode_last_error()
  rc = lib.mode_sphere_conv_fwd(one, one, one, one, one, 1, 4, 16, 32, 4, 3, 3, 2, 2, 16, 32, 1, null)
  assert rc == -1 and b'position table' in lib.mode_last_error()
  rc = lib.mode_sphere_conv_bwd_weight(one, one, one, one, null, 1, 4, 16, 32, 4, 3, 3, 1, 1, 16, 32, 1, null)
  assert rc == -3
  with pytest.raises(RuntimeError, match='code -1'):
    mode_hip.check(-1, 'x')


@pytest.mark.parametrize('typ,ih,iw,stride', [('ERP', 8, 16, 1), ('Cassini', 16, 8, 1), ('ERP', 8, 16, 2)])
def test_adjoint_table_is_the_transpose_of_the_gather(typ, ih, iw, stride):
  """mode_sphere_adjoint_build is host code: check it against the oracle's col2im (cu:293-356 restated)."""
  import numpy as np
  import torch
  from oracle import mode_ref, sphere_conv_ref
  pos = mode_ref.sphere_position(ih, iw, typ)
  H, W = pos.shape[2:]
  Ho, Wo = sphere_conv_ref.out_size(H, 3, stride, 1, 1), sphere_conv_ref.out_size(W, 3, stride, 1, 1)
  lib = mode_hip.lib()
  nmax = lib.mode_sphere_adjoint_max_entries(3, 3, Ho, Wo)
  rowptr = torch.empty(9 * H * W + 1, dtype=torch.int32)
  entries = torch.empty(2 * nmax, dtype=torch.int32)
  n = ctypes.c_int64(0)
  rc = lib.mode_sphere_adjoint_build(ctypes.c_void_p(pos.data_ptr()), H, W, 3, 3, stride, stride, Ho, Wo,
                                     ctypes.c_void_p(rowptr.data_ptr()), ctypes.c_void_p(entries.data_ptr()),
                                     ctypes.cast(ctypes.pointer(n), ctypes.c_void_p))
  assert rc == 0 and 0 < n.value <= nmax and int(rowptr[-1]) == n.value
  assert bool((rowptr[1:] >= rowptr[:-1]).all())
  ent = entries[:2 * n.value].view(-1, 2)
  p_idx = ent[:, 0].long()
  wts = ent[:, 1].contiguous().view(torch.float32).double()
  g = torch.Generator().manual_seed(0)
  gcol = torch.randn(1, 2, 9, Ho, Wo, generator=g, dtype=torch.float64)
  want = sphere_conv_ref.col2im_scatter(gcol, pos, H, W, 3, 3, stride, stride)  # (1,2,H,W)
  rows = torch.repeat_interleave(torch.arange(9 * H * W), (rowptr[1:] - rowptr[:-1]).long())
  k_idx, q_idx = rows // (H * W), rows % (H * W)
  got = torch.zeros(2, H * W, dtype=torch.float64)
  got.index_add_(1, q_idx, gcol[0].reshape(2, 9, Ho * Wo)[:, k_idx, p_idx] * wts)
  assert (got.view(2, H, W) - want[0]).abs().max() < 2e-6  # the table's weights are fp32 (kernel arithmetic), the oracle's fp64
  # rows are filled in ascending output-pixel order (deterministic summation)
  for r in (0, 5 * H * W + 3, 9 * H * W - 1):
    seg = p_idx[int(rowptr[r]):int(rowptr[r + 1])]
    assert bool((seg[1:] >= seg[:-1]).all())


def test_workspace_queries_are_host_only():
  lib = mode_hip.lib()
  n = lib.mode_sphere_conv_wpack_bytes(128, 128, 3, 3, 1)
  assert n >= 128 * 128 * 9 * 4 and n % 16 == 0
  assert lib.mode_sphere_conv_wpack_bytes(3, 4, 3, 3, 1) > 0
  ws = lib.mode_sphere_conv_bwd_weight_workspace_bytes(2, 128, 128, 3, 3, 256, 128, 1)
  assert ws > 0 and ws % (128 * 128 * 4) == 0
