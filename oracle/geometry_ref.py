"""CPU restatement of the export-stage geometry (TEST INFRASTRUCTURE ONLY -- imported by tests/ and nothing else).

Follows the reference line by line:
  disp2depth               save_output_disparity_stage.py:105-160
  cassini2equirec          utils/geometry.py:7-45
  rotate_cassini           utils/geometry.py:48-96
  depth_view_trans         utils/geometry.py:99-145 with the sequential z-buffer of :148-156 as a plain Python loop
  erp2rect_cassini         utils/geometry.py:160-198
Pinned against the imported reference by tests/golden/geometry.npz (tests/golden/make_golden_geometry.py: the reference module
itself, with a pass-through stand-in for the absent `numba.jit` decorator and an identity `.cuda()`) -- except `depth_left` /
`disp2depth`: the script holding it cannot be imported (argparse at import, torchvision and cv2 absent), so the reference FUNCTION
is taken out of it with ast and run by tests/golden/make_golden_disp2depth.py; tests/golden/disp2depth.npz pins this restatement
(tests/test_geometry.py: 5e-5 relative -- the reference evaluates in float64 under the container's NumPy 2, this file in float32)."""
import math

import numpy as np
import torch
import torch.nn.functional as F


def _nchw(img):
  a = np.asarray(img)
  if a.ndim == 2:
    a = a[:, :, None]
  return torch.FloatTensor(a).unsqueeze(0).transpose(1, 3).transpose(2, 3)


def _sample(src, gx, gy):
  grid = torch.cat([torch.FloatTensor(gx).unsqueeze(-1), torch.FloatTensor(gy).unsqueeze(-1)], dim=-1).unsqueeze(0)
  grid = grid.repeat_interleave(src.shape[0], dim=0)
  return F.grid_sample(src, grid, mode='bilinear', align_corners=True, padding_mode='border')


def rotation(pitch, yaw, roll):
  Rx = np.array([[1, 0, 0], [0, np.cos(roll), -np.sin(roll)], [0, np.sin(roll), np.cos(roll)]])
  Rz = np.array([[np.cos(yaw), -np.sin(yaw), 0], [np.sin(yaw), np.cos(yaw), 0], [0, 0, 1]])
  Ry = np.array([[np.cos(pitch), 0, -np.sin(pitch)], [0, 1, 0], [np.sin(pitch), 0, np.cos(pitch)]])
  return np.dot(np.dot(Rx, Rz), Ry)


def cassini_maps(h, w):
  theta_range = np.arange(np.pi - (np.pi / h), -np.pi, -(2 * np.pi / h))
  theta_map = np.array([theta_range for _ in range(w)]).astype(np.float32).T
  phi_range = np.arange(0.5 * np.pi - (0.5 * np.pi / w), -0.5 * np.pi, -(np.pi / w))
  phi_map = np.array([phi_range for _ in range(h)]).astype(np.float32)
  return theta_map, phi_map


def rotate_cassini(cassini_1, pitch, yaw, roll):
  R_I = np.linalg.inv(rotation(pitch, yaw, roll))
  h, w = cassini_1.shape[:2]
  theta_2_map, phi_2_map = cassini_maps(h, w)
  x_2 = np.sin(phi_2_map)
  y_2 = np.cos(phi_2_map) * np.sin(theta_2_map)
  z_2 = np.cos(phi_2_map) * np.cos(theta_2_map)
  X_1 = np.matmul(R_I, np.expand_dims(np.dstack((x_2, y_2, z_2)), axis=-1))
  theta_1_map = np.arctan2(X_1[:, :, 1, 0], X_1[:, :, 2, 0])
  phi_1_map = np.arcsin(np.clip(X_1[:, :, 0, 0], -1, 1))
  out = _sample(_nchw(cassini_1), np.clip(-phi_1_map / (0.5 * np.pi), -1, 1), np.clip(-theta_1_map / np.pi, -1, 1))
  return out.transpose(1, 3).transpose(1, 2).numpy()[0].astype(cassini_1.dtype)


def cassini2equirec(cassini):
  src = _nchw(cassini)
  erp_h, erp_w = src.shape[-1], src.shape[-2]
  theta_erp_range = np.arange(np.pi - (np.pi / erp_w), -np.pi, -(2 * np.pi / erp_w))
  theta_erp_map = np.array([theta_erp_range for _ in range(erp_h)]).astype(np.float32)
  phi_erp_range = np.arange(0.5 * np.pi - (0.5 * np.pi / erp_h), -0.5 * np.pi, -(np.pi / erp_h))
  phi_erp_map = np.array([phi_erp_range for _ in range(erp_w)]).astype(np.float32).T
  theta_cassini_map = np.arctan2(np.tan(phi_erp_map), np.cos(theta_erp_map))
  phi_cassini_map = np.arcsin(np.cos(phi_erp_map) * np.sin(theta_erp_map))
  out = _sample(src, np.clip(-phi_cassini_map / (0.5 * np.pi), -1, 1), np.clip(-theta_cassini_map / np.pi, -1, 1))
  return out.transpose(1, 3).transpose(1, 2).numpy()[0].astype(np.asarray(cassini).dtype).squeeze()


def erp2rect_cassini(erp, R, ca_h, ca_w):
  theta_ca_map, phi_ca_map = cassini_maps(ca_h, ca_w)
  x = np.sin(phi_ca_map)
  y = np.cos(phi_ca_map) * np.sin(theta_ca_map)
  z = np.cos(phi_ca_map) * np.cos(theta_ca_map)
  X2 = np.matmul(np.linalg.inv(R), np.expand_dims(np.dstack((x, y, z)), axis=-1))
  phi_erp_map = np.arcsin(X2[:, :, 1, :])
  theta_erp_map = np.arctan2(X2[:, :, 0, :], X2[:, :, 2, :])
  grid = torch.cat([torch.FloatTensor(np.clip(-theta_erp_map / np.pi, -1, 1)), torch.FloatTensor(np.clip(-phi_erp_map / (0.5 * np.pi), -1, 1))],
                   dim=-1).unsqueeze(0)
  out = F.grid_sample(_nchw(erp), grid, mode='bilinear', align_corners=True, padding_mode='border')
  return out.transpose(1, 3).transpose(1, 2).numpy()[0].astype(np.asarray(erp).dtype).squeeze()


def zbuffer(output_h, output_w, conf_1, conf_2, r_1, r_2, view_2, I_2, J_2):
  """geometry.py:148-156, the sequential scatter."""
  for i in range(output_h):
    for j in range(output_w):
      if r_1[i, j] > 0:
        flag = r_2[i, j] < view_2[I_2[i, j], J_2[i, j]]
        view_2[I_2[i, j], J_2[i, j]] = flag * r_2[i, j] + (1 - flag) * view_2[I_2[i, j], J_2[i, j]]
        conf_2[I_2[i, j], J_2[i, j]] = flag * conf_1[i, j] + (1 - flag) * conf_2[I_2[i, j], J_2[i, j]]
  return view_2, conf_2


def project(view_1, y0, z0, x0, pitch, yaw, roll):
  """geometry.py:100-137: r_2 (float64), target rows / columns, and the unrounded target coordinates."""
  R = rotation(pitch, yaw, roll)
  t = np.array([[x0], [y0], [z0]])
  h, w = view_1.shape
  theta_1_map, phi_1_map = cassini_maps(h, w)
  r_1 = view_1
  x_1 = r_1 * np.sin(phi_1_map)
  y_1 = r_1 * np.cos(phi_1_map) * np.sin(theta_1_map)
  z_1 = r_1 * np.cos(phi_1_map) * np.cos(theta_1_map)
  X_2 = np.matmul(R, np.expand_dims(np.dstack((x_1, y_1, z_1)), axis=-1) - t)
  r_2 = np.sqrt(np.square(X_2[:, :, 0, 0]) + np.square(X_2[:, :, 1, 0]) + np.square(X_2[:, :, 2, 0]))
  theta_2_map = np.arctan2(X_2[:, :, 1, 0], X_2[:, :, 2, 0])
  with np.errstate(invalid='ignore', divide='ignore'):
    phi_2_map = np.arcsin(np.clip(X_2[:, :, 0, 0] / r_2, -1, 1))
  fi = h / 2 - h * theta_2_map / (2 * np.pi)
  fj = w / 2 - w * phi_2_map / np.pi
  with np.errstate(invalid='ignore'):
    I_2 = np.clip(np.rint(fi), 0, h - 1).astype(np.int16)
    J_2 = np.clip(np.rint(fj), 0, w - 1).astype(np.int16)
  return r_2, I_2, J_2, fi, fj


def depth_view_trans(view_1, conf_1, y0, z0, x0, pitch, yaw, roll):
  h, w = view_1.shape
  r_2, I_2, J_2, _, _ = project(view_1, y0, z0, x0, pitch, yaw, roll)
  view_2 = np.ones((h, w)).astype(np.float32) * 100000
  conf_2 = np.zeros((h, w)).astype(np.float32)
  view_2, conf_2 = zbuffer(h, w, conf_1, conf_2, view_1, r_2, view_2, I_2, J_2)
  view_2[view_2 == 100000] = 0
  view_2 = view_2.astype(np.float32)
  view_2[view_2 > 1000] = 1000
  return view_2, conf_2


def baselines(dbname):
  if dbname == 'Deep360':
    return np.array([1, 1, math.sqrt(2), math.sqrt(2), 1, 1]).astype(np.float32)
  return np.array([0.6 * math.sqrt(2), 0.6 * math.sqrt(2), 1.2, 1.2, 0.6 * math.sqrt(2), 0.6 * math.sqrt(2)]).astype(np.float32)


def depth_left(disp, baseline):
  """save_output_disparity_stage.py:118-135: sine-rule depth in the left camera's frame.  The Python-float constants are
  written as float32 scalars: under the value-based casting of the NumPy 1.x the reference was written for, `float32 masked
  array * math.pi` stays float32, whereas NumPy 2's masked arrays promote it to float64 -- the float32 reading is the
  reference's (and the kernel's)."""
  h, w = disp.shape
  disp = np.asarray(disp, dtype=np.float32)
  phi_l_range = np.arange(0.5 * math.pi - (0.5 * math.pi / w), -0.5 * math.pi, -(math.pi / w))
  phi_l_map = np.array([phi_l_range for _ in range(h)]).astype(np.float32)
  disp_not_0 = np.ma.array(disp, mask=disp == 0)
  phi_r_map = disp_not_0 * np.float32(math.pi) / np.float32(w) + phi_l_map
  depth_l = np.float32(baseline) * np.sin(np.float32(math.pi / 2) - phi_r_map) / np.sin(phi_r_map - phi_l_map)
  depth_l = depth_l.filled(1000)
  depth_l[depth_l > 1000] = 1000
  depth_l[depth_l < 0] = 0
  return depth_l


def disp2depth(disp, conf_map, cam_pair, dbname='Deep360'):
  pair = {'12': 0, '13': 1, '14': 2, '23': 3, '24': 4, '34': 5}[cam_pair]
  depth_l = depth_left(disp, baselines(dbname)[pair])
  if cam_pair == '12':
    return depth_l, conf_map
  if cam_pair in ('13', '14'):
    a = 0.5 * math.pi if cam_pair == '13' else 0.25 * math.pi
    return rotate_cassini(depth_l[:, :, None], a, 0, 0)[:, :, 0], rotate_cassini(conf_map[:, :, None], a, 0, 0)[:, :, 0]
  if cam_pair == '23':
    return depth_view_trans(depth_l, conf_map, 0, -math.sqrt(2) / 2, -math.sqrt(2) / 2, 0.75 * math.pi, 0, 0)
  if cam_pair == '24':
    return depth_view_trans(depth_l, conf_map, 0, -1, 0, 0.5 * math.pi, 0, 0)
  return depth_view_trans(depth_l, conf_map, 0, 1, 0, 0, 0, 0)
