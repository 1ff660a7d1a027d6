"""Export-stage geometry of the disparity network, MI355X edition.

Drop-in for the functions of the reference's ``utils/geometry.py`` (cassini2Equirec, rotateCassini, depthViewTransWithConf,
erp2rect_cassini) and for ``disp2depth`` of ``save_output_disparity_stage.py:105-160``: same names, arguments and return
conventions (numpy in, numpy out).  The reference builds angle maps with numpy, ships them to the GPU for one
``F.grid_sample`` and back, and runs the z-buffer of the view transform as a sequential numba loop on the CPU.  Here the
angle maps (functions of the image size and the rotation only) are still built with numpy -- they are what defines the
geometry -- but are cached and kept on the device, and everything per pixel runs in libmode_hip.so
(csrc/geometry.hip): ``*_gpu`` variants take and return device tensors so that a pipeline never leaves the GPU.

There is no CPU path: the native library is required (the reference, too, hard-codes ``.cuda()``).
"""
import functools
import math

import numpy as np
import torch

from mode_hip import check, lib, ptr, require_f32c, require_gpu, stream_of

_DEV = 'cuda'


def _ranges(output_h, output_w):
  """theta over the h axis (longitude, 2 pi) and phi over the w axis (latitude, pi) of a Cassini image, as float64 ranges
  (geometry.py:64-74): np.arange(start, end, -step)."""
  theta = np.arange(np.pi - (np.pi / output_h), -np.pi, -(2 * np.pi / output_h))
  phi = np.arange(0.5 * np.pi - (0.5 * np.pi / output_w), -0.5 * np.pi, -(np.pi / output_w))
  return theta, phi


def _cassini_angle_maps(output_h, output_w):
  """float32 (H, W) maps theta[i], phi[j] exactly as the reference builds them (list of ranges -> float32 array)."""
  theta, phi = _ranges(output_h, output_w)
  theta_map = np.broadcast_to(theta.astype(np.float32)[:, None], (output_h, output_w))
  phi_map = np.broadcast_to(phi.astype(np.float32)[None, :], (output_h, output_w))
  return theta_map, phi_map


def _rotation(pitch, yaw, roll):
  """R = Rx(roll) Rz(yaw) Ry(pitch), geometry.py:49-55."""
  Rx = np.array([[1, 0, 0], [0, np.cos(roll), -np.sin(roll)], [0, np.sin(roll), np.cos(roll)]])
  Rz = np.array([[np.cos(yaw), -np.sin(yaw), 0], [np.sin(yaw), np.cos(yaw), 0], [0, 0, 1]])
  Ry = np.array([[np.cos(pitch), 0, -np.sin(pitch)], [0, 1, 0], [np.sin(pitch), 0, np.cos(pitch)]])
  return np.dot(np.dot(Rx, Rz), Ry)


def _unit_dirs(output_h, output_w):
  """(3, H, W) float32: sin(phi), cos(phi) sin(theta), cos(phi) cos(theta) (geometry.py:76-78, 126-128 without the radius)."""
  theta_map, phi_map = _cassini_angle_maps(output_h, output_w)
  return np.stack((np.sin(phi_map), np.cos(phi_map) * np.sin(theta_map), np.cos(phi_map) * np.cos(theta_map))).astype(np.float32)


@functools.lru_cache(maxsize=32)
def _trig_device(output_h, output_w, device):
  """sin phi[W], cos phi[W], sin theta[H], cos theta[H] as numpy's float32 sin / cos of the float32 angle ranges."""
  theta, phi = _ranges(output_h, output_w)
  theta, phi = theta.astype(np.float32), phi.astype(np.float32)
  return torch.from_numpy(np.concatenate((np.sin(phi), np.cos(phi), np.sin(theta), np.cos(theta))).astype(np.float32)).to(device)


def _grid_sample(src, grid):
  """src (N,C,Hs,Ws), grid (1|N,Ho,Wo,2) device tensors -> (N,C,Ho,Wo); bilinear, border padding, align_corners=True."""
  require_gpu(src, grid)
  src, grid = src.contiguous(), grid.contiguous()
  require_f32c(src, grid)
  N, C, Hs, Ws = src.shape
  G, Ho, Wo, two = grid.shape
  if two != 2 or G not in (1, N):
    raise RuntimeError('grid_sample: grid %s does not fit input %s' % (tuple(grid.shape), tuple(src.shape)))
  dst = torch.empty((N, C, Ho, Wo), dtype=src.dtype, device=src.device)
  with torch.cuda.device_of(src):
    check(lib().mode_grid_sample_border(ptr(src), ptr(grid), ptr(dst), N, C, Hs, Ws, Ho, Wo, G, stream_of(src)), 'mode_grid_sample_border')
  return dst


def _to_nchw(img):
  """numpy (H, W) or (H, W, C) -> device (1, C, H, W) float32, as `torch.FloatTensor(img).unsqueeze(0).transpose(1, 3).transpose(2, 3)`."""
  a = np.asarray(img)
  if a.ndim == 2:
    a = a[:, :, None]
  return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)).astype(np.float32)).unsqueeze(0).to(_DEV)


def _from_nchw(sampled, like):
  return sampled[0].permute(1, 2, 0).cpu().numpy().astype(np.asarray(like).dtype)


# ------------------------------------------------------------------------------------------------ rotateCassini
@functools.lru_cache(maxsize=32)
def _rotate_grid(output_h, output_w, pitch, yaw, roll, device):
  R_I = np.linalg.inv(_rotation(pitch, yaw, roll))
  dirs = _unit_dirs(output_h, output_w)
  X_2 = np.expand_dims(np.dstack((dirs[0], dirs[1], dirs[2])), axis=-1)
  X_1 = np.matmul(R_I, X_2)
  theta_1_map = np.arctan2(X_1[:, :, 1, 0], X_1[:, :, 2, 0])
  phi_1_map = np.arcsin(np.clip(X_1[:, :, 0, 0], -1, 1))
  grid = np.stack((np.clip(-phi_1_map / (0.5 * np.pi), -1, 1), np.clip(-theta_1_map / np.pi, -1, 1)), axis=-1).astype(np.float32)
  return torch.from_numpy(grid).unsqueeze(0).to(device)


def rotateCassini_gpu(src, pitch, yaw, roll):
  """src (N, C, H, W) device tensor -> the same views rotated (geometry.py:48-96)."""
  return _grid_sample(src, _rotate_grid(src.shape[2], src.shape[3], float(pitch), float(yaw), float(roll), str(src.device)))


def rotateCassini(cassini_1, pitch, yaw, roll):
  """numpy (H, W, C) -> numpy (H, W, C) of the same dtype."""
  return _from_nchw(rotateCassini_gpu(_to_nchw(cassini_1), pitch, yaw, roll), cassini_1)


# ------------------------------------------------------------------------------------------------ cassini2Equirec
@functools.lru_cache(maxsize=32)
def _c2e_grid(erp_h, erp_w, device):
  theta_erp = np.arange(np.pi - (np.pi / erp_w), -np.pi, -(2 * np.pi / erp_w))
  phi_erp = np.arange(0.5 * np.pi - (0.5 * np.pi / erp_h), -0.5 * np.pi, -(np.pi / erp_h))
  theta_erp_map = np.broadcast_to(theta_erp.astype(np.float32)[None, :], (erp_h, erp_w))
  phi_erp_map = np.broadcast_to(phi_erp.astype(np.float32)[:, None], (erp_h, erp_w))
  theta_cassini_map = np.arctan2(np.tan(phi_erp_map), np.cos(theta_erp_map))
  phi_cassini_map = np.arcsin(np.cos(phi_erp_map) * np.sin(theta_erp_map))
  grid = np.stack((np.clip(-phi_cassini_map / (0.5 * np.pi), -1, 1), np.clip(-theta_cassini_map / np.pi, -1, 1)), axis=-1).astype(np.float32)
  return torch.from_numpy(grid).unsqueeze(0).to(device)


def cassini2Equirec(cassini):
  """geometry.py:7-45: Cassini (H=2h, W=h) image(s) -> equirectangular (h, 2h).  numpy (H,W) / (H,W,C) -> numpy, squeezed;
  a 4D device tensor (N,C,H,W) -> device tensor (N,C,h,2h) squeezed on dim 1, as the reference does."""
  if isinstance(cassini, torch.Tensor) and cassini.dim() == 4:
    src = cassini
    return _grid_sample(src, _c2e_grid(src.shape[-1], src.shape[-2], str(src.device))).squeeze(1)
  a = np.asarray(cassini)
  if a.ndim not in (2, 3):
    raise ValueError('cassini2Equirec: expected a 2D/3D array or a 4D tensor')
  src = _to_nchw(a)
  out = _grid_sample(src, _c2e_grid(src.shape[-1], src.shape[-2], str(src.device)))
  return _from_nchw(out, a).squeeze()


# ------------------------------------------------------------------------------------------------ erp2rect_cassini
def erp2rect_cassini(erp, R, ca_h, ca_w, devcice='cuda'):
  """geometry.py:160-198: equirectangular image -> rectified Cassini (ca_h, ca_w) under rotation R (3x3).  The misspelt
  keyword is the reference's."""
  dirs = _unit_dirs(ca_h, ca_w)
  X = np.expand_dims(np.dstack((dirs[0], dirs[1], dirs[2])), axis=-1)
  X2 = np.matmul(np.linalg.inv(R), X)
  phi_erp_map = np.arcsin(X2[:, :, 1, :])
  theta_erp_map = np.arctan2(X2[:, :, 0, :], X2[:, :, 2, :])
  grid = np.concatenate((np.clip(-theta_erp_map / np.pi, -1, 1), np.clip(-phi_erp_map / (0.5 * np.pi), -1, 1)), axis=-1).astype(np.float32)
  grid = torch.from_numpy(grid).unsqueeze(0).to(devcice)
  if isinstance(erp, torch.Tensor) and erp.dim() == 4:
    return _grid_sample(erp, grid).squeeze(1)
  a = np.asarray(erp)
  out = _grid_sample(_to_nchw(a).to(devcice), grid)
  return _from_nchw(out, a).squeeze()


# ------------------------------------------------------------------------------------------------ depthViewTransWithConf
def depthViewTransWithConf_gpu(view_1, conf_1, y0, z0, x0, pitch, yaw, roll):
  """view_1, conf_1: (H, W) float32 device tensors -> (view_2, conf_2) device tensors (geometry.py:99-156)."""
  require_gpu(view_1, conf_1)
  view_1, conf_1 = view_1.contiguous(), conf_1.contiguous()
  require_f32c(view_1, conf_1)
  H, W = view_1.shape
  R = np.ascontiguousarray(_rotation(pitch, yaw, roll), dtype=np.float64)
  t = np.array([x0, y0, z0], dtype=np.float64)
  trig = _trig_device(H, W, str(view_1.device))
  view_2, conf_2 = torch.empty_like(view_1), torch.empty_like(view_1)
  ws = torch.empty(lib().mode_depth_view_trans_workspace_bytes(H, W) // 8, dtype=torch.int64, device=view_1.device)
  with torch.cuda.device_of(view_1):
    check(lib().mode_depth_view_trans(ptr(view_1), ptr(conf_1), ptr(trig), R.ctypes.data, t.ctypes.data, ptr(view_2), ptr(conf_2), ptr(ws),
                                      H, W, stream_of(view_1)), 'mode_depth_view_trans')
  return view_2, conf_2


def project_gpu(view_1, y0, z0, x0, pitch, yaw, roll):
  """First half of depthViewTransWithConf_gpu: (r2 float64 (H, W), target index int32 (H, W), -1 = source takes no part)."""
  require_gpu(view_1)
  view_1 = view_1.contiguous()
  require_f32c(view_1)
  H, W = view_1.shape
  R = np.ascontiguousarray(_rotation(pitch, yaw, roll), dtype=np.float64)
  t = np.array([x0, y0, z0], dtype=np.float64)
  r2 = torch.empty((H, W), dtype=torch.float64, device=view_1.device)
  tgt = torch.empty((H, W), dtype=torch.int32, device=view_1.device)
  with torch.cuda.device_of(view_1):
    check(lib().mode_depth_view_project(ptr(view_1), ptr(_trig_device(H, W, str(view_1.device))), R.ctypes.data, t.ctypes.data, ptr(r2),
                                        ptr(tgt), H, W, stream_of(view_1)), 'mode_depth_view_project')
  return r2, tgt


def zbuffer_gpu(r2, target, conf_1):
  """Second half: the reference's z-buffer over (r2 float64, target int32, conf float32) triples of any shape -> (view_2, conf_2)."""
  require_gpu(r2, target, conf_1)
  r2, target, conf_1 = r2.contiguous(), target.contiguous(), conf_1.contiguous()
  n = r2.numel()
  view_2 = torch.empty(r2.shape, dtype=torch.float32, device=r2.device)
  conf_2 = torch.empty_like(view_2)
  ws = torch.empty(n, dtype=torch.int64, device=r2.device)
  with torch.cuda.device_of(r2):
    check(lib().mode_zbuffer(ptr(r2), ptr(target), ptr(conf_1), ptr(view_2), ptr(conf_2), ptr(ws), n, stream_of(r2)), 'mode_zbuffer')
  return view_2, conf_2


def depthViewTransWithConf(view_1, conf_1, y0, z0, x0, pitch, yaw, roll):
  """numpy (H, W) depth and confidence seen from camera 1 -> the same scene seen from a camera at (x0, y0, z0) rotated by
  (pitch, yaw, roll); float32 numpy out."""
  v = torch.from_numpy(np.ascontiguousarray(view_1, dtype=np.float32)).to(_DEV)
  c = torch.from_numpy(np.ascontiguousarray(conf_1, dtype=np.float32)).to(_DEV)
  v2, c2 = depthViewTransWithConf_gpu(v, c, y0, z0, x0, pitch, yaw, roll)
  return v2.cpu().numpy(), c2.cpu().numpy()


# ------------------------------------------------------------------------------------------------ disp2depth
CAM_PAIRS = {'12': 0, '13': 1, '14': 2, '23': 3, '24': 4, '34': 5}


def _baselines(dbname):
  """save_output_disparity_stage.py:108-113 (3D60 defines none)."""
  if dbname == 'Deep360':
    return np.array([1, 1, math.sqrt(2), math.sqrt(2), 1, 1]).astype(np.float32)
  if dbname == '3D60':
    raise ValueError('disp2depth: the reference defines no baselines for 3D60')
  return np.array([0.6 * math.sqrt(2), 0.6 * math.sqrt(2), 1.2, 1.2, 0.6 * math.sqrt(2), 0.6 * math.sqrt(2)]).astype(np.float32)


def disp2depth_gpu(disp, conf_map, cam_pair, dbname='Deep360'):
  """disp, conf_map: (H, W) float32 device tensors -> (depth, conf) in the reference frame of camera 1, device tensors."""
  if cam_pair not in CAM_PAIRS:
    print("Error! Wrong Cam_pair!")
    return None
  require_gpu(disp, conf_map)
  disp = disp.contiguous()
  require_f32c(disp)
  H, W = disp.shape
  depth_l = torch.empty_like(disp)
  with torch.cuda.device_of(disp):
    check(lib().mode_disp2depth(ptr(disp), ptr(depth_l), H, W, float(_baselines(dbname)[CAM_PAIRS[cam_pair]]), stream_of(disp)), 'mode_disp2depth')
  if cam_pair == '12':
    return depth_l, conf_map
  if cam_pair in ('13', '14'):
    angle = 0.5 * math.pi if cam_pair == '13' else 0.25 * math.pi
    both = rotateCassini_gpu(torch.stack((depth_l, conf_map.to(depth_l.dtype))).unsqueeze(0), angle, 0, 0)[0]
    return both[0], both[1]
  if cam_pair == '23':
    return depthViewTransWithConf_gpu(depth_l, conf_map, 0, -math.sqrt(2) / 2, -math.sqrt(2) / 2, 0.75 * math.pi, 0, 0)
  if cam_pair == '24':
    return depthViewTransWithConf_gpu(depth_l, conf_map, 0, -1, 0, 0.5 * math.pi, 0, 0)
  return depthViewTransWithConf_gpu(depth_l, conf_map, 0, 1, 0, 0, 0, 0)


def disp2depth(disp, conf_map, cam_pair, dbname='Deep360'):
  """numpy in, numpy out (save_output_disparity_stage.py:105-160; `dbname` replaces the script's global args.dbname)."""
  d = torch.from_numpy(np.ascontiguousarray(disp, dtype=np.float32)).to(_DEV)
  c = torch.from_numpy(np.ascontiguousarray(conf_map, dtype=np.float32)).to(_DEV)
  out = disp2depth_gpu(d, c, cam_pair, dbname)
  if out is None:
    return None
  return out[0].cpu().numpy(), out[1].cpu().numpy()
