"""ctypes binding of libmode_hip.so (C-ABI declared in include/mode_hip.h).

This is the only place that touches the native library.  There is NO fallback: if the library is
missing or a tensor is not on a GPU, the call raises.  PyTorch is used for device memory and
streams only (tensor.data_ptr(), torch.cuda.current_stream()).
"""
import ctypes
import os
import threading

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'libmode_hip.so')

_c_int = ctypes.c_int
_c_ptr = ctypes.c_void_p
_c_size = ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/mode_hip.h one to one (checked by tests/test_abi.py)
SIGNATURES = {
    'mode_hip_abi_version': (_c_int, []),
    'mode_last_error': (ctypes.c_char_p, []),
    'mode_debug_poison': (_c_int, [ctypes.c_uint, _c_ptr]),
    'mode_weight_pack_reuse': (_c_int, [_c_int]),
    'mode_sum_n': (_c_int, [_c_ptr] * 5 + [ctypes.c_longlong, _c_ptr]),
    'mode_conv3d_fwd_split_stats_partials': (_c_int, []),
    'mode_conv3d_fwd_split_stats': (_c_int, [_c_ptr] * 5 + [_c_int] * 6 + [_c_ptr]),
    'mode_bn_train_fwd_prestats': (_c_int, [_c_ptr] * 7 + [ctypes.c_float] * 2 + [_c_int] + [_c_ptr] * 6 + [_c_int] * 3 +
                                   [ctypes.c_longlong, _c_ptr]),
    'mode_sphere_conv_wpack_bytes': (_c_size, [_c_int] * 5),
    'mode_sphere_conv_fwd': (_c_int, [_c_ptr] * 5 + [_c_int] * 12 + [_c_ptr]),
    'mode_sphere_conv_bwd_data': (_c_int, [_c_ptr] * 5 + [_c_int] * 12 + [_c_ptr]),
    'mode_sphere_adjoint_max_entries': (_c_size, [_c_int] * 4),
    'mode_sphere_adjoint_build': (_c_int, [_c_ptr] + [_c_int] * 8 + [_c_ptr] * 3),
    'mode_sphere_conv_bwd_data_adj': (_c_int, [_c_ptr] * 6 + [_c_int] * 11 + [_c_ptr]),
    'mode_sphere_conv_bwd_data_adj_list': (_c_int, [_c_ptr] * 6 + [_c_int] * 11 + [_c_ptr, _c_int, _c_ptr]),
    'mode_sphere_adjplan_build': (_c_int, [_c_ptr] + [_c_int] * 4 + [_c_ptr] * 7),
    'mode_sphere_conv_bwd_data_win_wpack_bytes': (_c_size, [_c_int] * 5),
    'mode_sphere_conv_bwd_data_win_supported': (_c_int, [_c_int] * 3),
    'mode_sphere_conv_bwd_data_win_split': (_c_int, [_c_ptr] * 5 + [_c_int] + [_c_ptr] * 4 + [_c_int] * 9 + [_c_ptr]),
    'mode_sphere_conv_bwd_data_win_split_f16': (_c_int, [_c_ptr] * 7 + [_c_int] + [_c_ptr] * 4 + [_c_int] * 9 + [_c_ptr]),
    'mode_sphere_plan_max_tiles': (_c_size, [_c_int] * 2),
    'mode_sphere_plan_build': (_c_int, [_c_ptr] + [_c_int] * 4 + [_c_ptr] * 2),
    'mode_sphere_conv_win_wpack_bytes': (_c_size, [_c_int] * 5),
    'mode_sphere_conv_fwd_win': (_c_int, [_c_ptr] * 6 + [_c_int] * 12 + [_c_ptr]),
    'mode_transpose_planes': (_c_int, [_c_ptr] * 2 + [ctypes.c_longlong] + [_c_int] * 2 + [_c_ptr]),
    'mode_sphere_plan_rest_pixels': (_c_int, [_c_ptr] * 2 + [_c_int] * 2 + [_c_ptr] * 2),
    'mode_sphere_plan_polar_max_items': (_c_size, [_c_ptr]),
    'mode_sphere_plan_polar': (_c_int, [_c_ptr] * 3 + [_c_int] * 2 + [_c_ptr] * 4),
    'mode_sphere_conv_bwd_weight_win_workspace_bytes': (_c_size, [_c_int] * 11),
    'mode_sphere_plan_records_count': (_c_size, [_c_int]),
    'mode_sphere_plan_records': (_c_int, [_c_ptr] * 3 + [_c_int] * 2 + [_c_ptr] * 2),
    'mode_sphere_conv_bwd_weight_win': (_c_int, [_c_ptr] * 6 + [_c_int] * 3 + [_c_ptr] * 3 + [_c_int] + [_c_ptr] * 3 + [_c_int] * 9 +
                                        [_c_ptr] * 3),
    'mode_sphere_conv_bwd_weight_win_split': (_c_int, [_c_ptr] * 6 + [_c_int] * 3 + [_c_ptr] * 3 + [_c_int] + [_c_ptr] * 3 + [_c_int] * 9 +
                                              [_c_ptr] * 3),
    'mode_sphere_conv_bwd_weight_win_split_f16': (_c_int, [_c_ptr] * 8 + [_c_int] * 3 + [_c_ptr] * 3 + [_c_int] + [_c_ptr] * 3 + [_c_int] * 9 +
                                                  [_c_ptr] * 3),
    'mode_sphere_conv_bwd_weight_workspace_bytes': (_c_size, [_c_int] * 8),
    'mode_sphere_conv_bwd_weight': (_c_int, [_c_ptr] * 5 + [_c_int] * 12 + [_c_ptr]),
    'mode_sphere_conv_fwd_bn': (_c_int, [_c_ptr] * 6 + [_c_int] * 12 + [_c_ptr]),
    'mode_sphere_conv_fwd_win_bn': (_c_int, [_c_ptr] * 7 + [_c_int] * 12 + [_c_ptr]),
    'mode_sphere_conv_fwd_win_split': (_c_int, [_c_ptr] * 7 + [_c_int] * 12 + [_c_ptr]),
    'mode_sphere_conv_fwd_win_split_f16': (_c_int, [_c_ptr] * 8 + [_c_int] * 12 + [_c_ptr]),
    'mode_cost_volume_fwd': (_c_int, [_c_ptr] * 3 + [_c_int] * 5 + [_c_ptr]),
    'mode_cost_volume_bwd': (_c_int, [_c_ptr] * 3 + [_c_int] * 5 + [_c_ptr]),
    'mode_conv2d_wpack_bytes': (_c_size, [_c_int] * 2),
    'mode_conv2d_fwd': (_c_int, [_c_ptr] * 4 + [_c_int] * 6 + [_c_ptr]),
    'mode_conv2d_fwd_bn': (_c_int, [_c_ptr] * 5 + [_c_int] * 6 + [_c_ptr]),
    'mode_zero_insert2': (_c_int, [_c_ptr] * 2 + [ctypes.c_longlong] + [_c_int] * 2 + [_c_ptr]),
    'mode_conv2d_bwd_data': (_c_int, [_c_ptr] * 4 + [_c_int] * 6 + [_c_ptr]),
    'mode_conv2d_bwd_weight_workspace_bytes': (_c_size, [_c_int] * 5),
    'mode_conv2d_bwd_weight': (_c_int, [_c_ptr] * 4 + [_c_int] * 7 + [_c_ptr]),
    'mode_conv1x1_wpack_bytes': (_c_size, [_c_int] * 2),
    'mode_conv1x1_fwd': (_c_int, [_c_ptr] * 4 + [_c_int] * 6 + [_c_ptr]),
    'mode_conv1x1_fwd_bn': (_c_int, [_c_ptr] * 5 + [_c_int] * 6 + [_c_ptr]),
    'mode_conv1x1_bwd_data': (_c_int, [_c_ptr] * 4 + [_c_int] * 6 + [_c_ptr]),
    'mode_conv1x1_bwd_weight_workspace_bytes': (_c_size, [_c_int] * 6),
    'mode_conv1x1_bwd_weight': (_c_int, [_c_ptr] * 4 + [_c_int] * 7 + [_c_ptr]),
    'mode_conv_stem_wpack_bytes': (_c_size, [_c_int] * 2),
    'mode_conv_stem_fwd': (_c_int, [_c_ptr] * 4 + [_c_int] * 5 + [_c_ptr]),
    'mode_conv_stem_fwd_bn': (_c_int, [_c_ptr] * 5 + [_c_int] * 5 + [_c_ptr]),
    'mode_conv_stem_bwd_weight_workspace_bytes': (_c_size, [_c_int] * 5),
    'mode_conv_stem_bwd_weight': (_c_int, [_c_ptr] * 4 + [_c_int] * 6 + [_c_ptr]),
    'mode_cost_conv_assemble_fwd': (_c_int, [_c_ptr] * 3 + [_c_int] * 5 + [_c_ptr]),
    'mode_cost_conv_assemble_fwd_bn': (_c_int, [_c_ptr] * 4 + [_c_int] * 5 + [_c_ptr]),
    'mode_cost_conv_assemble_fwd_bn_amax': (_c_int, [_c_ptr] * 5 + [_c_int] * 5 + [_c_ptr]),
    'mode_cost_conv_assemble_bwd': (_c_int, [_c_ptr] * 3 + [_c_int] * 5 + [_c_ptr]),
    'mode_conv3d_wpack_bytes': (_c_size, [_c_int] * 2),
    'mode_conv3d_fwd': (_c_int, [_c_ptr] * 4 + [_c_int] * 7 + [_c_ptr]),
    'mode_conv3d_fwd_bn': (_c_int, [_c_ptr] * 5 + [_c_int] * 7 + [_c_ptr]),
    'mode_deconv3d_fwd_bn': (_c_int, [_c_ptr] * 5 + [_c_int] * 6 + [_c_ptr]),
    'mode_conv3d_bwd_data': (_c_int, [_c_ptr] * 4 + [_c_int] * 7 + [_c_ptr]),
    'mode_conv2d_split_supported': (_c_int, [_c_int] * 4),
    'mode_conv2d_fwd_split': (_c_int, [_c_ptr] * 5 + [_c_int] * 6 + [_c_ptr]),
    'mode_conv2d_bwd_data_split': (_c_int, [_c_ptr] * 4 + [_c_int] * 6 + [_c_ptr]),
    'mode_conv2d_bwd_data_split_acc': (_c_int, [_c_ptr] * 5 + [_c_int] * 6 + [_c_ptr]),
    'mode_conv2d_fwd_split_f16': (_c_int, [_c_ptr] * 6 + [_c_int] * 6 + [_c_ptr]),
    'mode_conv2d_fwd_split_f16_bn': (_c_int, [_c_ptr] * 7 + [_c_int] * 6 + [_c_ptr]),
    'mode_conv2d_bwd_data_split_f16': (_c_int, [_c_ptr] * 7 + [_c_int] * 6 + [_c_ptr]),
    'mode_conv2d_bwd_weight_split': (_c_int, [_c_ptr] * 4 + [_c_int] * 7 + [_c_ptr]),
    'mode_conv2d_bwd_weight_split_f16': (_c_int, [_c_ptr] * 6 + [_c_int] * 7 + [_c_ptr]),
    'mode_conv3d_split_supported': (_c_int, [_c_int] * 4),
    'mode_conv3d_fwd_split': (_c_int, [_c_ptr] * 5 + [_c_int] * 6 + [_c_ptr]),
    'mode_conv3d_fwd_s2_split': (_c_int, [_c_ptr] * 5 + [_c_int] * 6 + [_c_ptr]),
    'mode_conv3d_fwd_s2_split_amax': (_c_int, [_c_ptr] * 6 + [_c_int] * 6 + [_c_ptr]),
    'mode_deconv3d_split_supported': (_c_int, [_c_int] * 2),
    'mode_deconv3d_fwd_split': (_c_int, [_c_ptr] * 4 + [_c_int] * 6 + [_c_ptr]),
    'mode_deconv3d_split_bn_supported': (_c_int, [_c_int] * 2),
    'mode_conv3d_bwd_data_split_acc_supported': (_c_int, [_c_int] * 3),
    'mode_abs_max': (_c_int, [_c_ptr, ctypes.c_longlong, _c_ptr, _c_ptr]),
    'mode_abs_max_batch': (_c_int, [_c_ptr, _c_ptr, _c_int, _c_ptr, _c_ptr]),
    'mode_conv3d_fwd_split_f16': (_c_int, [_c_ptr] * 6 + [_c_int] * 6 + [_c_ptr]),
    'mode_conv3d_fwd_split_f16_bn': (_c_int, [_c_ptr] * 7 + [_c_int] * 6 + [_c_ptr]),
    'mode_conv3d_bwd_data_split_f16': (_c_int, [_c_ptr] * 7 + [_c_int] * 6 + [_c_ptr]),
    'mode_conv3d_bwd_weight_split_f16': (_c_int, [_c_ptr] * 6 + [_c_int] * 7 + [_c_ptr]),
    'mode_conv3d_bwd_data_split_acc': (_c_int, [_c_ptr] * 5 + [_c_int] * 7 + [_c_ptr]),
    'mode_deconv3d_fwd_split_bn': (_c_int, [_c_ptr] * 5 + [_c_int] * 6 + [_c_ptr]),
    'mode_deconv3d_fwd_split_bn_amax': (_c_int, [_c_ptr] * 6 + [_c_int] * 6 + [_c_ptr]),
    'mode_conv3d_bwd_data_s2_split': (_c_int, [_c_ptr] * 4 + [_c_int] * 6 + [_c_ptr]),
    'mode_conv3d_bwd_data_split': (_c_int, [_c_ptr] * 4 + [_c_int] * 6 + [_c_ptr]),
    'mode_conv3d_bwd_weight_split': (_c_int, [_c_ptr] * 4 + [_c_int] * 7 + [_c_ptr]),
    'mode_conv3d_bwd_weight_s2_split': (_c_int, [_c_ptr] * 4 + [_c_int] * 7 + [_c_ptr]),
    'mode_conv3d_bwd_weight_workspace_bytes': (_c_size, [_c_int] * 7),
    'mode_deconv3d_fwd': (_c_int, [_c_ptr] * 4 + [_c_int] * 6 + [_c_ptr]),
    'mode_conv3d_bwd_weight': (_c_int, [_c_ptr] * 4 + [_c_int] * 8 + [_c_ptr]),
    'mode_head_fwd': (_c_int, [_c_ptr] * 3 + [_c_int] * 7 + [_c_ptr]),
    'mode_head_bwd_workspace_bytes': (_c_size, [_c_int] * 4),
    'mode_head_bwd': (_c_int, [_c_ptr] * 4 + [_c_int] * 7 + [_c_ptr]),
    'mode_smooth_l1_workspace_bytes': (_c_size, [ctypes.c_longlong]),
    'mode_smooth_l1_masked': (_c_int, [_c_ptr] * 4 + [ctypes.c_float] * 3 + [_c_ptr] * 3 + [ctypes.c_longlong, _c_ptr]),
    'mode_head_loss_supported': (_c_int, [_c_int] * 7),
    'mode_head_bwd_loss': (_c_int, [_c_ptr] * 3 + [ctypes.c_float] + [_c_ptr] * 3 + [_c_int] * 7 + [_c_ptr]),
    'mode_disp2depth': (_c_int, [_c_ptr] * 2 + [_c_int] * 2 + [ctypes.c_float, _c_ptr]),
    'mode_grid_sample_border': (_c_int, [_c_ptr] * 3 + [_c_int] * 7 + [_c_ptr]),
    'mode_depth_view_trans_workspace_bytes': (_c_size, [_c_int] * 2),
    'mode_depth_view_trans': (_c_int, [_c_ptr] * 8 + [_c_int] * 2 + [_c_ptr]),
    'mode_depth_view_project': (_c_int, [_c_ptr] * 6 + [_c_int] * 2 + [_c_ptr]),
    'mode_zbuffer': (_c_int, [_c_ptr] * 6 + [ctypes.c_longlong, _c_ptr]),
    'mode_classif_workspace_bytes': (_c_size, [_c_int] * 5),
    'mode_classif_train_fwd': (_c_int, [_c_ptr] * 6 + [ctypes.c_float] * 2 + [_c_ptr] * 8 + [_c_int] * 5 + [_c_ptr]),
    'mode_classif_train_bwd': (_c_int, [_c_ptr] * 13 + [_c_int] + [_c_ptr] + [_c_int] * 5 + [_c_ptr]),
    'mode_bn_workspace_bytes': (_c_size, [_c_int]),
    'mode_bn_train_fwd': (_c_int, [_c_ptr] * 7 + [ctypes.c_float] * 2 + [_c_int] + [_c_ptr] * 6 + [_c_int] * 2 +
                          [ctypes.c_longlong, _c_int, _c_ptr]),
    'mode_bn_eval_fwd': (_c_int, [_c_ptr] * 6 + [ctypes.c_float, _c_int] + [_c_ptr] + [_c_int] * 2 + [ctypes.c_longlong, _c_ptr]),
    'mode_bn_train_bwd': (_c_int, [_c_ptr] * 8 + [_c_int] + [_c_ptr] * 4 + [_c_int] + [_c_ptr] + [_c_int] * 2 +
                          [ctypes.c_longlong, _c_int, _c_ptr]),
    # the `_amax` entries: the plain entry's arguments + the maximum's buffer in front of the stream (ABI 30)
    'mode_bn_train_fwd_amax': (_c_int, [_c_ptr] * 7 + [ctypes.c_float] * 2 + [_c_int] + [_c_ptr] * 6 + [_c_int] * 2 +
                               [ctypes.c_longlong, _c_int, _c_ptr, _c_ptr]),
    'mode_bn_train_fwd_prestats_amax': (_c_int, [_c_ptr] * 7 + [ctypes.c_float] * 2 + [_c_int] + [_c_ptr] * 6 + [_c_int] * 3 +
                                        [ctypes.c_longlong, _c_ptr, _c_ptr]),
    'mode_bn_train_bwd_amax': (_c_int, [_c_ptr] * 8 + [_c_int] + [_c_ptr] * 4 + [_c_int] + [_c_ptr] + [_c_int] * 2 +
                               [ctypes.c_longlong, _c_int, _c_ptr, _c_ptr]),
    'mode_classif_train_bwd_amax': (_c_int, [_c_ptr] * 13 + [_c_int] + [_c_ptr] + [_c_int] * 5 + [_c_ptr, _c_ptr]),
    # fusion network (csrc/fusion_ops.hip)
    'mode_maxpool2x2_fwd': (_c_int, [_c_ptr] * 2 + [ctypes.c_longlong, _c_int, _c_int, _c_ptr]),
    'mode_maxpool2x2_bwd': (_c_int, [_c_ptr] * 3 + [ctypes.c_longlong, _c_int, _c_int, _c_ptr]),
    'mode_depth_to_space2': (_c_int, [_c_ptr] * 4 + [_c_int] * 5 + [_c_ptr]),
    'mode_space_to_depth2': (_c_int, [_c_ptr] * 2 + [_c_int] * 4 + [_c_ptr]),
    'mode_conv1x1_sigmoid_fwd': (_c_int, [_c_ptr] * 4 + [_c_int] * 2 + [ctypes.c_longlong, _c_ptr]),
    'mode_conv1x1_sigmoid_bwd_workspace_bytes': (_c_size, [_c_int, ctypes.c_longlong]),
    'mode_conv1x1_sigmoid_bwd': (_c_int, [_c_ptr] * 7 + [_c_int] + [_c_ptr] + [_c_int] * 2 + [ctypes.c_longlong, _c_ptr]),
}

ABI_VERSION = 31  # MODE_HIP_ABI_VERSION of include/mode_hip.h this binding was written against
_lib = None
_lock = threading.Lock()


def lib():
  """Load (once) and return the native library; raises if it has not been built."""
  global _lib
  if _lib is None:
    with _lock:
      if _lib is None:
        if not os.path.exists(LIB_PATH):
          raise RuntimeError('libmode_hip.so is not built (%s missing): run `python -c "import __graft_entry__ as g; '
                             'g.build()"` or `python mode-2022_amd/mode_hip/build.py`' % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
          fn = getattr(handle, name)
          fn.restype = res
          fn.argtypes = args
        have = handle.mode_hip_abi_version()
        if have != ABI_VERSION:
          raise RuntimeError('libmode_hip.so has ABI version %d, this binding needs %d: rebuild it (python '
                             'mode-2022_amd/mode_hip/build.py)' % (have, ABI_VERSION))
        _lib = handle
  return _lib


class BnEpilogue(ctypes.Structure):
  """struct mode_bn_epilogue of include/mode_hip.h (host-side struct of device pointers)."""
  _fields_ = [('gamma', _c_ptr), ('beta', _c_ptr), ('mean', _c_ptr), ('var', _c_ptr), ('eps', ctypes.c_float), ('add', _c_ptr),
              ('relu', _c_int)]


def check(rc, what):
  if rc != 0:
    msg = lib().mode_last_error()
    raise RuntimeError('%s failed (code %d): %s' % (what, rc, msg.decode() if msg else ''))


def ptr(t):
  return ctypes.c_void_p(t.data_ptr())


def stream_of(t):
  return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def require_gpu(*tensors):
  """Same refusal as the reference op (sphere_conv.py:33-34): no CPU path exists."""
  for t in tensors:
    if t is not None and not t.is_cuda:
      raise NotImplementedError('Only support cuda tensor!')


def require_f32c(*tensors):
  for t in tensors:
    if t.dtype != torch.float32:
      raise TypeError('libmode_hip kernels are fp32 (got %s)' % t.dtype)
    if not t.is_contiguous():
      raise ValueError('libmode_hip kernels need contiguous tensors')
