"""Export-stage geometry: oracle vs the reference's golden vectors (CPU), HIP kernels vs both (GPU)."""
import math

import numpy as np
import pytest
import torch

from oracle import geometry_ref as G

TRANSFORMS = ('t23', 't24', 't34', 'tid', 'tgen')


# ----------------------------------------------------------------------------------------------------------- CPU tier
@pytest.mark.parametrize('tag', ['a', 'b', 'c'])
def test_oracle_view_transform_is_the_reference(golden, tag):
  """depth_view_trans incl. the sequential z-buffer: bit-identical to the imported reference."""
  z = golden('geometry.npz')
  for name in TRANSFORMS:
    v, c = G.depth_view_trans(z[tag + '/depth'].copy(), z[tag + '/conf'].copy(), *z['%s/%s/args' % (tag, name)].tolist())
    assert np.array_equal(v, z['%s/%s/view' % (tag, name)]), (tag, name)
    assert np.array_equal(c, z['%s/%s/conf' % (tag, name)]), (tag, name)


@pytest.mark.parametrize('tag', ['a', 'b', 'c'])
def test_oracle_reprojections_are_the_reference(golden, tag):
  z = golden('geometry.npz')
  img = z[tag + '/img']
  assert np.array_equal(G.rotate_cassini(img, 0.5 * np.pi, 0, 0), z[tag + '/rot13'])
  assert np.array_equal(G.rotate_cassini(img, 0.3, -0.4, 1.1), z[tag + '/rot_gen'])
  if tag + '/c2e' in z.files:
    assert np.array_equal(G.cassini2equirec(img), z[tag + '/c2e'])
    assert np.array_equal(G.erp2rect_cassini(z[tag + '/erp'], z[tag + '/e2c_R'], img.shape[0], img.shape[1]), z[tag + '/e2c'])


def test_oracle_disp2depth_properties():
  """The sine rule is not pinned by the reference (see oracle/geometry_ref.py): check what it must satisfy."""
  rng = np.random.RandomState(3)
  disp = (rng.rand(64, 32).astype(np.float32) * 20)
  disp[::7, ::5] = 0
  d = G.depth_left(disp, np.float32(1.0))
  assert d.dtype == np.float32 and (d[disp == 0] == 1000).all() and (d >= 0).all() and (d <= 1000).all()
  # depth of a point at angular disparity delta seen at latitude phi_l over baseline b: b * cos(phi_l + delta) / sin(delta)
  j = np.arange(32)
  phi_l = (0.5 * math.pi - 0.5 * math.pi / 32 - j * math.pi / 32)[None, :]
  delta = disp.astype(np.float64) * math.pi / 32
  ok = (disp != 0)
  with np.errstate(divide='ignore', invalid='ignore'):
    want = np.clip(np.cos(phi_l + delta) / np.sin(delta), 0, 1000)
  assert np.allclose(d[ok], want[ok], rtol=2e-4, atol=1e-4)
  # a larger baseline scales the depth
  assert np.allclose(G.depth_left(disp, np.float32(2.0))[ok & (d < 400)], 2 * d[ok & (d < 400)], rtol=1e-5)


# ----------------------------------------------------------------------------------------------------------- GPU tier
DEV = 'cuda:0'


def _ambiguous(fi, fj, eps=1e-6):
  """Sources whose unrounded target coordinate sits on a rounding boundary (x.5): there the last bit of atan2 / asin decides
  the pixel.  The reference's maps put whole families of points exactly there (a pure rotation by a multiple of the angular
  step maps pixel centres onto pixel EDGES: H/2 - H*theta/(2 pi) = i + 0.5), so this is not a corner case."""
  with np.errstate(invalid='ignore'):
    return (np.abs(np.abs(fi - np.floor(fi)) - 0.5) < eps) | (np.abs(np.abs(fj - np.floor(fj)) - 0.5) < eps) | ~np.isfinite(fi) | ~np.isfinite(fj)


@pytest.mark.gpu
@pytest.mark.parametrize('tag', ['a', 'b', 'c'])
def test_hip_zbuffer_bit_exact(golden, tag):
  """The atomic-min z-buffer reproduces the reference's sequential loop bit for bit (ties, duplicates, holes, the 100000
  sentinel) when it is given the same (r2, target) pairs; checked against the reference's own outputs."""
  from utils import geometry as HG
  z = golden('geometry.npz')
  depth, conf = z[tag + '/depth'], z[tag + '/conf']
  h, w = depth.shape
  for name in TRANSFORMS:
    r2, I, J, _, _ = G.project(depth, *z['%s/%s/args' % (tag, name)].tolist())
    tgt = np.where(depth > 0, I.astype(np.int32) * w + J.astype(np.int32), -1).astype(np.int32)
    v, c = HG.zbuffer_gpu(torch.from_numpy(np.nan_to_num(r2, nan=1e9)).to(DEV), torch.from_numpy(tgt).to(DEV), torch.from_numpy(conf).to(DEV))
    assert np.array_equal(v.cpu().numpy(), z['%s/%s/view' % (tag, name)]), (tag, name)
    assert np.array_equal(c.cpu().numpy(), z['%s/%s/conf' % (tag, name)]), (tag, name)


@pytest.mark.gpu
@pytest.mark.parametrize('tag', ['a', 'b'])
def test_hip_fused_view_transform_generic_pose_is_the_reference(golden, tag):
  """A generic pose has no systematic rounding ties: the fused kernel matches the reference's output outright."""
  from utils import geometry as HG
  z = golden('geometry.npz')
  v, c = HG.depthViewTransWithConf(z[tag + '/depth'], z[tag + '/conf'], *z[tag + '/tgen/args'].tolist())
  bad = (v != z[tag + '/tgen/view']) | (c != z[tag + '/tgen/conf'])
  assert bad.sum() <= 1, int(bad.sum())


@pytest.mark.gpu
def test_hip_zbuffer_adversarial_ties():
  """Many sources per target, radii that are equal in float32 but not in float64, radii that round up to the sentinel."""
  from utils import geometry as HG
  rng = np.random.RandomState(2)
  n, nt = 20000, 64
  base = np.float32(rng.rand(n) * 3 + 1).astype(np.float64)
  r2 = base * (1 + rng.randint(-3, 4, n) * 2.0**-27)  # clusters of float64 values around the same float32
  r2[rng.rand(n) < 0.02] = 99999.999  # below 100000 in float64, rounds to 100000.0 in float32
  r2[rng.rand(n) < 0.02] = 100000.0
  r1 = np.where(rng.rand(n) < 0.1, 0, 1).astype(np.float32)
  tgt = rng.randint(0, nt, n).astype(np.int32)
  conf = rng.rand(n).astype(np.float32)
  view_2 = np.ones((1, n), np.float32) * 100000
  conf_2 = np.zeros((1, n), np.float32)
  G.zbuffer(1, n, conf[None], conf_2, r1[None], r2[None], view_2, np.zeros((1, n), np.int64), tgt[None].astype(np.int64))
  view_2[view_2 == 100000] = 0
  view_2[view_2 > 1000] = 1000
  v, c = HG.zbuffer_gpu(torch.from_numpy(r2).to(DEV), torch.from_numpy(np.where(r1 > 0, tgt, -1).astype(np.int32)).to(DEV),
                        torch.from_numpy(conf).to(DEV))
  assert np.array_equal(v.cpu().numpy(), view_2[0]) and np.array_equal(c.cpu().numpy(), conf_2[0])


@pytest.mark.gpu
@pytest.mark.parametrize('tag', ['a', 'b', 'c'])
def test_hip_projection(golden, tag):
  """r2 to float64 round-off; the target pixel identical wherever it is well defined."""
  from utils import geometry as HG
  z = golden('geometry.npz')
  depth = z[tag + '/depth']
  h, w = depth.shape
  for name in TRANSFORMS:
    args = z['%s/%s/args' % (tag, name)].tolist()
    r2_ref, I, J, fi, fj = G.project(depth, *args)
    r2, tgt = HG.project_gpu(torch.from_numpy(depth).to(DEV), *args)
    r2, tgt = r2.cpu().numpy(), tgt.cpu().numpy()
    live = (depth > 0) & (r2_ref < 100000) & (r2_ref > 0)
    assert np.array_equal(tgt >= 0, live)
    assert np.abs(r2[live] - r2_ref[live]).max() <= 1e-12 * r2_ref[live].max()
    clear = live & ~_ambiguous(fi, fj)
    assert np.array_equal(tgt[clear], (I.astype(np.int32) * w + J)[clear]), (tag, name)
    # on a boundary the pixel may go either way, but never further than the two candidates
    di = np.abs(tgt // w - I)[live & ~clear]
    dj = np.abs(tgt % w - J)[live & ~clear]
    assert (di <= 1).all() and (dj <= 1).all()


@pytest.mark.gpu
def test_hip_view_transform_full_size():
  """1024x512, generic pose: the fused kernel against the literal sequential loop; deterministic despite the atomics."""
  from utils import geometry as HG
  g = torch.Generator().manual_seed(5)
  H, W = 1024, 512
  depth = torch.rand(H, W, generator=g) * 9 + 0.5
  depth[torch.rand(H, W, generator=g) < 0.1] = 0
  conf = torch.rand(H, W, generator=g)
  d, c = depth.to(DEV), conf.to(DEV)
  pose = (0.1, -1, 0.2, 0.5 * math.pi + 0.013, 0.3, -0.2)
  v1, k1 = HG.depthViewTransWithConf_gpu(d, c, *pose)
  v2, k2 = HG.depthViewTransWithConf_gpu(d, c, *pose)
  assert torch.equal(v1, v2) and torch.equal(k1, k2)
  rv, rc = G.depth_view_trans(depth.numpy(), conf.numpy(), *pose)
  _, _, _, fi, fj = G.project(depth.numpy(), *pose)
  n_amb = int((_ambiguous(fi, fj, 1e-9) & (depth.numpy() > 0)).sum())
  bad = (v1.cpu().numpy() != rv) | (k1.cpu().numpy() != rc)
  assert bad.sum() <= 2 * n_amb + 2, (int(bad.sum()), n_amb)
  # fused == project + zbuffer
  r2, tgt = HG.project_gpu(d, *pose)
  v3, k3 = HG.zbuffer_gpu(r2, tgt, c)
  assert torch.equal(v1, v3) and torch.equal(k1, k3)


@pytest.mark.gpu
@pytest.mark.parametrize('tag', ['a', 'b', 'c'])
def test_hip_reprojections(golden, tag):
  from utils import geometry as HG
  z = golden('geometry.npz')
  img = z[tag + '/img']
  tol = 2e-6  # float32 bilinear weights: same formula as ATen's, a different rounding order at most
  assert np.abs(HG.rotateCassini(img, 0.5 * np.pi, 0, 0) - z[tag + '/rot13']).max() < tol
  assert np.abs(HG.rotateCassini(img, 0.3, -0.4, 1.1) - z[tag + '/rot_gen']).max() < tol
  if tag + '/c2e' in z.files:
    assert np.abs(HG.cassini2Equirec(img) - z[tag + '/c2e']).max() < tol
    assert np.abs(HG.erp2rect_cassini(z[tag + '/erp'], z[tag + '/e2c_R'], img.shape[0], img.shape[1]) - z[tag + '/e2c']).max() < tol
    t4 = torch.from_numpy(img.transpose(2, 0, 1)).unsqueeze(0).to(DEV)  # tensor in, tensor out (geometry.py:15, 43-45)
    e = HG.cassini2Equirec(t4)
    assert e.shape == (1, 3, img.shape[1], img.shape[0]) and np.abs(e[0].permute(1, 2, 0).cpu().numpy() - z[tag + '/c2e']).max() < tol


@pytest.mark.gpu
@pytest.mark.parametrize('dbname', ['Deep360', 'other'])
@pytest.mark.parametrize('pair', ['12', '13', '14', '23', '24', '34'])
def test_hip_disp2depth(pair, dbname):
  from utils import geometry as HG
  rng = np.random.RandomState(11)
  disp = (rng.rand(128, 64).astype(np.float32) * 20)
  disp[rng.rand(128, 64) < 0.1] = 0
  conf = rng.rand(128, 64).astype(np.float32)
  d, c = HG.disp2depth(disp, conf, pair, dbname)
  rd, rc = G.disp2depth(disp, conf, pair, dbname)
  if pair == '12':
    far = rd >= 999
    assert np.allclose(d[~far], rd[~far], rtol=2e-5, atol=1e-5) and np.array_equal(c, conf)
    assert (np.abs(d[far] - rd[far]) < 1).all()
  elif pair in ('13', '14'):
    # bilinear resampling of a depth map with 1000-valued holes: compare where the neighbourhood is smooth
    assert np.median(np.abs(d - rd)) < 1e-4 and np.abs(c - rc).max() < 1e-5
  else:
    # the poses of the camera pairs are rotations by multiples of pi/4 about the longitude axis: far points land exactly on
    # pixel edges (see _ambiguous), where the last bit of atan2 decides -- compare the well-defined part: the filled fraction
    # and the distribution of depths
    assert abs((d > 0).mean() - (rd > 0).mean()) < 0.05
    assert np.allclose(np.sort(d[d > 0])[::37][:20], np.sort(rd[rd > 0])[::37][:20], rtol=0.05)
  assert HG.disp2depth(disp, conf, '99') is None


@pytest.mark.parametrize('dbname', ['Deep360', 'other'])
@pytest.mark.parametrize('pair', ['12', '13', '14', '23', '24', '34'])
def test_oracle_disp2depth_pinned_by_the_reference_function(golden, pair, dbname):
  """tests/golden/disp2depth.npz holds outputs of the reference's OWN disp2depth (save_output_disparity_stage.py:105-160, taken
  out of the un-importable script with ast by tests/golden/make_golden_disp2depth.py).  Under the container's NumPy 2 the
  reference evaluates the sine rule in float64 for the direct pairs; the oracle keeps float32 throughout: agreement to 5e-5
  relative away from the 1000 m clip, the clipped / masked pixels identical."""
  z = golden('disp2depth.npz')
  d, c = G.disp2depth(z['disp'].copy(), z['conf'].copy(), pair, dbname)
  rd, rc = z['%s/%s/depth' % (dbname, pair)], z['%s/%s/conf' % (dbname, pair)]
  far = rd >= 999
  assert (np.abs(d - rd)[~far] <= 1e-4 * np.maximum(np.abs(rd[~far]), 1e-3)).all()
  assert (np.abs(d[far] - rd[far]) < 1).all()
  assert np.abs(c - rc).max() < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize('dbname', ['Deep360', 'other'])
def test_hip_disp2depth_against_the_reference_function(golden, dbname):
  """The HIP sine-rule kernel (mode_disp2depth) against the reference function's own output, direct pair '12'."""
  from utils import geometry as HG
  z = golden('disp2depth.npz')
  d, c = HG.disp2depth(z['disp'].copy(), z['conf'].copy(), '12', dbname)
  rd = z['%s/12/depth' % dbname]
  far = rd >= 999
  assert (np.abs(d - rd)[~far] <= 1e-4 * np.maximum(np.abs(rd[~far]), 1e-3)).all()
  assert (np.abs(d[far] - rd[far]) < 1).all() and np.array_equal(c, z['conf'])
