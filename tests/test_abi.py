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


def test_workspace_queries_are_host_only():
  lib = mode_hip.lib()
  n = lib.mode_sphere_conv_wpack_bytes(128, 128, 3, 3, 1)
  assert n >= 128 * 128 * 9 * 4 and n % 16 == 0
  assert lib.mode_sphere_conv_wpack_bytes(3, 4, 3, 3, 1) > 0
  ws = lib.mode_sphere_conv_bwd_weight_workspace_bytes(2, 128, 128, 3, 3, 256, 128, 1)
  assert ws > 0 and ws % (128 * 128 * 4) == 0
