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
  assert mode_hip.lib().mode_hip_abi_version() == mode_hip.ABI_VERSION == 31


def test_maximum_buffer_size_matches_the_header():
  """functional.BN_ABSMAX_FLOATS (what Python allocates for mode_abs_max / the `_amax` BatchNorm entries) is the header's MODE_BN_ABSMAX_FLOATS (what the
  kernels zero and read: 16 + 128 slots x 16 floats); the fp16 entries reject a missing maximum on the host."""
  from mode_hip import functional as HF
  src = open(os.path.join(ROOT, 'include', 'mode_hip.h')).read()
  m = re.search(r'#define\s+MODE_BN_ABSMAX_FLOATS\s+(\d+)', src)
  assert m and int(m.group(1)) == HF.BN_ABSMAX_FLOATS == 16 * (1 + 128)
  lib = mode_hip.lib()
  null, one = ctypes.c_void_p(0), ctypes.c_void_p(16)
  assert lib.mode_conv2d_fwd_split_f16(one, one, null, one, one, one, 1, 16, 8, 8, 16, 1, null) == -1 and b'maximum' in lib.mode_last_error()
  assert lib.mode_conv2d_bwd_weight_split_f16(one, one, one, null, one, one, 1, 16, 8, 8, 16, 1, 0, null) == -1 and b'maximum' in lib.mode_last_error()


def test_inference_entries_on_fp16_pieces_validate_on_the_host():
  """ABI 31 (mode_conv3d_fwd_split_f16_bn, mode_conv2d_fwd_split_f16_bn and the `_amax` eval entries): a missing epilogue or maximum buffer
  is refused before any launch, with a message."""
  from mode_hip import BnEpilogue
  lib = mode_hip.lib()
  null, one = ctypes.c_void_p(0), ctypes.c_void_p(16)
  e = BnEpilogue(one, one, one, one, 1e-5, None, 1)
  args3 = (1, 32, 4, 8, 32, 32, null)
  assert lib.mode_conv3d_fwd_split_f16_bn(one, one, one, None, one, one, one, *args3) == -1 and b'BatchNorm' in lib.mode_last_error()
  assert lib.mode_conv3d_fwd_split_f16_bn(one, one, null, ctypes.byref(e), one, one, one, *args3) == -1 and b'maximum' in lib.mode_last_error()
  assert lib.mode_conv3d_fwd_split_f16_bn(one, one, one, ctypes.byref(e), one, null, one, *args3) == -1 and b'maximum' in lib.mode_last_error()
  args2 = (1, 32, 8, 32, 32, 1, null)
  assert lib.mode_conv2d_fwd_split_f16_bn(one, one, one, None, one, one, one, *args2) == -1 and b'BatchNorm' in lib.mode_last_error()
  assert lib.mode_conv2d_fwd_split_f16_bn(one, one, null, ctypes.byref(e), one, one, one, *args2) == -1 and b'maximum' in lib.mode_last_error()
  # the output maximum of the stride-2 entry comes out of the eval epilogue: without one there is nothing to fill it
  assert lib.mode_conv3d_fwd_s2_split_amax(one, one, None, one, one, one, 1, 32, 4, 8, 32, 64, null) == -1 and b'maximum' in lib.mode_last_error()


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


def test_sphere_window_plan_covers_every_sample():
  """mode_sphere_plan_build (host code, no GPU): every tile of the gnomonic Cassini table gets a window class, the classes
  partition the tiles, and every live sample of a tile lies inside the window the plan gives it."""
  import numpy as np
  import torch
  from oracle import mode_ref
  lib = mode_hip.lib()
  for ih, iw in ((128, 256), (10, 20), (33, 66)):
    pos = mode_ref.sphere_position(ih, iw, 'Cassini').contiguous()
    H, W = pos.shape[2:]
    n = lib.mode_sphere_plan_max_tiles(H, W)
    assert n == -(-H // 64) * -(-W // 4)
    tiles = torch.full((4 * n,), -1, dtype=torch.int32)
    counts = torch.zeros(4, dtype=torch.int32)
    assert lib.mode_sphere_plan_build(mode_hip.ptr(pos), H, W, 3, 3, mode_hip.ptr(tiles), mode_hip.ptr(counts)) == 0
    c = counts.tolist()
    assert sum(c) == n and c[3] == 0, c
    t = tiles.view(n, 4).numpy()
    assert len({(a, b) for a, b, _, _ in t}) == n  # every tile exactly once
    p = pos[0].numpy()
    assert [int((t[:, 3] >> 16 == k).sum()) for k in range(4)] == c
    for h0, w0, rbase, packed in t:
      cbase, wr = packed & 0xffff, {0: 81, 1: 145, 2: H + 1}[packed >> 16]
      hh, ww = np.meshgrid(np.arange(h0, min(h0 + 64, H)), np.arange(w0, min(w0 + 4, W)), indexing='ij')
      for k in range(9):
        y, x = p[2 * k][hh, ww], p[2 * k + 1][hh, ww]
        live = (y > -1) & (x > -1) & (y < H) & (x < W)
        r0 = np.maximum(np.floor(y).astype(int), 0)[live]
        c0 = np.maximum(np.floor(x).astype(int), 0)[live]
        lr = (r0 - rbase) % H
        assert (lr + 1 < wr).all() and (c0 - cbase >= 0).all() and (c0 - cbase + 1 < 8).all()
    if (ih, iw) == (128, 256):
      assert c[0] == 7 * (c[1] + c[2])  # 28 of the 32 column blocks are far enough from the poles for the small window


def test_sphere_window_plan_rejects_unstructured_table():
  import torch
  lib = mode_hip.lib()
  g = torch.Generator().manual_seed(3)
  H, W = 24, 20
  pos = torch.stack([torch.rand(9, H, W, generator=g) * (H + 1) - 1, torch.rand(9, H, W, generator=g) * (W + 1) - 1], 1).reshape(1, 18, H, W).contiguous()
  n = lib.mode_sphere_plan_max_tiles(H, W)
  tiles = torch.zeros(4 * n, dtype=torch.int32)
  counts = torch.zeros(4, dtype=torch.int32)
  assert lib.mode_sphere_plan_build(mode_hip.ptr(pos), H, W, 3, 3, mode_hip.ptr(tiles), mode_hip.ptr(counts)) == 0
  assert counts[3] > 0 and int(counts.sum()) == n


def test_sphere_polar_plan_items():
  """mode_sphere_plan_polar (host code): the tiles outside the small-window class become column items whose nine per-tap
  windows (34 rows x 2 columns) hold every live sample, and the records point inside them."""
  import numpy as np
  import torch
  from oracle import mode_ref
  lib = mode_hip.lib()
  pos = mode_ref.sphere_position(128, 256, 'Cassini').contiguous()
  H, W = pos.shape[2:]
  n = lib.mode_sphere_plan_max_tiles(H, W)
  tiles = torch.zeros(4 * n, dtype=torch.int32)
  counts = torch.zeros(4, dtype=torch.int32)
  assert lib.mode_sphere_plan_build(mode_hip.ptr(pos), H, W, 3, 3, mode_hip.ptr(tiles), mode_hip.ptr(counts)) == 0
  npmax = lib.mode_sphere_plan_polar_max_items(mode_hip.ptr(counts))
  assert npmax == (int(counts[1]) + int(counts[2])) * 8
  items = torch.zeros(20 * npmax, dtype=torch.int32)
  rw = torch.zeros(4 * 288 * npmax, dtype=torch.float32)
  ro = torch.zeros(288 * npmax, dtype=torch.int32)
  ni = torch.zeros(1, dtype=torch.int32)
  assert lib.mode_sphere_plan_polar(mode_hip.ptr(pos), mode_hip.ptr(tiles), mode_hip.ptr(counts), H, W, mode_hip.ptr(items), mode_hip.ptr(rw),
                                    mode_hip.ptr(ro), mode_hip.ptr(ni)) == 0
  ni = int(ni)
  assert ni == npmax  # 16 columns x 8 row blocks, all plannable
  it = items.view(-1, 20)[:ni].numpy()
  cols = sorted(set(it[:, 1].tolist()))
  assert cols == list(range(0, 8)) + list(range(W - 8, W))
  off = ro.view(-1, 9, 32)[:ni].numpy()
  wts = rw.view(-1, 9, 32, 4)[:ni].numpy()
  assert (off >= 0).all() and (off + 34 + 1 < 9 * 68 + 1).all()
  assert ((off // 68) == np.arange(9)[None, :, None])[wts.any(-1)].all()  # a live record points into its own tap's window
  p = pos[0].numpy()
  for i in (0, ni // 2, ni - 1):  # spot-check the window bases against the table
    h0, w = it[i, 0], it[i, 1]
    for k in range(9):
      y = p[2 * k][h0:h0 + 32, w]
      r0 = np.maximum(np.floor(y).astype(int), 0)
      lr = (r0 - it[i, 2 + k]) % H
      assert (lr + 1 < 34).all()


def test_bn_train_fwd_refuses_in_place_operation():
  """The normalisation pass re-reads the pivot of the shifted sums from y while it writes out: out == y is rejected on the host."""
  lib = mode_hip.lib()
  a, b = ctypes.c_void_p(4096), ctypes.c_void_p(8192)
  rc = lib.mode_bn_train_fwd(a, None, b, b, None, None, None, 0.1, 1e-5, 1, a, b, b, None, None, b, 2, 4, 64, 1, None)
  assert rc == -1 and b'in-place' in lib.mode_last_error()


def test_no_state_between_calls_in_the_abi():
  """the maximum of a tensor a BatchNorm pass writes is requested through a PARAMETER of the `_amax` entries, not through a
  one-shot setter in front of the plain call -- the header's own convention (no mutable state between calls) holds for every entry."""
  lib = mode_hip.lib()
  for gone in ('mode_bn_next_out_absmax', 'mode_bn_next_gy_absmax'):
    assert not hasattr(lib, gone), gone
  src = open(os.path.join(ROOT, 'mode-2022_amd', 'csrc', 'bn_act.hip')).read() + open(os.path.join(ROOT, 'mode-2022_amd', 'csrc', 'classif_head.hip')).read()
  assert 'thread_local' not in src
  # the `_amax` entries check their arguments like the plain ones (host-side, no launch): in-place is refused with a maximum buffer too
  a, b = ctypes.c_void_p(4096), ctypes.c_void_p(8192)
  rc = lib.mode_bn_train_fwd_amax(a, None, b, b, None, None, None, 0.1, 1e-5, 1, a, b, b, None, None, b, 2, 4, 64, 1, b, None)
  assert rc == -1 and b'in-place' in lib.mode_last_error()
  rc = lib.mode_bn_train_bwd_amax(None, a, None, b, b, b, None, None, 0, a, None, b, b, 0, b, 2, 4, 64, 1, b, None)
  assert rc == -1 and b'null pointer' in lib.mode_last_error()


def test_table_caches_are_bounded():
  from mode_hip import functional as HF
  c = HF._LRU(3)
  for i in range(5):
    c[i] = i * i
  assert len(c) == 3 and 0 not in c and 1 not in c and c[4] == 16
  assert c.get(2) == 4  # touching an entry makes it the newest
  c[9] = 81
  assert 2 in c and 3 not in c
  for i in range(HF.TABLE_CACHE_ENTRIES + 5):  # the integer tables of many resolutions do not accumulate
    HF.conv2d_table(8 + i, 8, 3, 3, (1, 1), (1, 1), (1, 1), 'cpu')
  assert len(HF._conv_tables) == HF.TABLE_CACHE_ENTRIES


def test_table_caches_never_evict_what_a_graph_capture_was_handed(monkeypatch):
  """A captured hipGraph holds the raw addresses of the tables its kernels were launched with: an entry that is looked
  up or stored while the stream is capturing is pinned -- it survives any number of later insertions -- until release_graph_pins()."""
  from mode_hip import functional as HF
  c = HF._LRU(3)
  state = {'capturing': False}
  monkeypatch.setattr(HF._LRU, 'capturing', staticmethod(lambda: state['capturing']))
  c['a'], c['b'] = object(), object()
  a = c['a']
  state['capturing'] = True
  assert c['a'] is a and c.get('b') is not None  # handed out during the capture -> pinned
  c['built-in-capture'] = 1
  state['capturing'] = False
  for i in range(10):
    c[i] = i
  assert c['a'] is a and 'b' in c and c['built-in-capture'] == 1 and len(c) == 3 + 3
  assert 0 not in c and 9 in c  # the unpinned part is still a 3-entry LRU
  c['a'] = a  # re-storing a pinned key keeps it pinned
  assert 'a' in c.pinned
  c.release_graph_pins()
  assert len(c) == 3 and not c.pinned


@pytest.mark.parametrize('ih,iw', [(128, 256), (32, 64)])
def test_adjoint_window_plan_reproduces_the_adjoint_table(ih, iw):
  """mode_sphere_adjplan_build (host code): on its good tiles the 4-slot records, read back through the window geometry, are exactly
  the (source pixel, weight) lists of mode_sphere_adjoint_build, in the same order; good and bad tiles partition the image."""
  import numpy as np
  import torch
  from oracle import mode_ref
  lib = mode_hip.lib()
  pos = mode_ref.sphere_position(ih, iw, 'Cassini').contiguous()
  H, W = pos.shape[2:]
  n = lib.mode_sphere_plan_max_tiles(H, W)
  good, bad, counts = torch.zeros(4 * n, dtype=torch.int32), torch.zeros(2 * n, dtype=torch.int32), torch.zeros(2, dtype=torch.int32)
  rec_off, rec_w = torch.zeros(n * 9 * 256 * 4, dtype=torch.int32), torch.zeros(n * 9 * 256 * 4, dtype=torch.float32)
  rec_off2, rec_w2 = torch.zeros(n * 9 * 256 * 2, dtype=torch.int32), torch.zeros(n * 9 * 256 * 2, dtype=torch.float32)
  assert lib.mode_sphere_adjplan_build(mode_hip.ptr(pos), H, W, 3, 3, mode_hip.ptr(good), mode_hip.ptr(bad), mode_hip.ptr(counts),
                                       mode_hip.ptr(rec_off), mode_hip.ptr(rec_w), mode_hip.ptr(rec_off2), mode_hip.ptr(rec_w2)) == 0
  ng, nb = counts.tolist()
  assert ng > 0 and ng + nb == n
  g, b = good[:4 * ng].view(ng, 4).numpy(), bad[:2 * nb].view(nb, 2).numpy()
  assert len({(int(a), int(c)) for a, c in g[:, :2]} | {(int(a), int(c)) for a, c in b}) == n
  nmax = lib.mode_sphere_adjoint_max_entries(3, 3, H, W)
  rowptr, entries, ne = torch.empty(9 * H * W + 1, dtype=torch.int32), torch.empty(2 * nmax, dtype=torch.int32), ctypes.c_int64(0)
  assert lib.mode_sphere_adjoint_build(mode_hip.ptr(pos), H, W, 3, 3, 1, 1, H, W, mode_hip.ptr(rowptr), mode_hip.ptr(entries),
                                       ctypes.cast(ctypes.pointer(ne), ctypes.c_void_p)) == 0
  rp, ent = rowptr.numpy(), entries[:2 * ne.value].view(-1, 2).numpy()
  ro = np.concatenate((rec_off[:ng * 9 * 256 * 4].view(ng, 9, 256, 4).numpy(), rec_off2[:ng * 9 * 256 * 2].view(ng, 9, 256, 2).numpy()), -1)
  rw = np.concatenate((rec_w[:ng * 9 * 256 * 4].view(ng, 9, 256, 4).numpy(), rec_w2[:ng * 9 * 256 * 2].view(ng, 9, 256, 2).numpy()), -1)
  assert (ro >= 0).all() and (ro < 8 * 81).all()
  six = (g[:, 3] >> 16) == 1
  assert ((ro[~six][..., 4:] == 0) & (rw[~six][..., 4:] == 0)).all()  # the 4-slot class never uses slots 4, 5
  if (ih, iw) == (128, 256):
    assert sorted(set(g[six][:, 1].tolist())) == [60, 64] and ng == 112  # the equator columns 63..65; all but the 16 polar tiles planned
  for i in [0, ng // 2, ng - 1] + np.nonzero(six)[0][:2].tolist():
    h0, w0, rbase, cbase = [int(v) for v in g[i]]
    cbase &= 0xffff
    for k in range(9):
      for pix in (0, 31, 100, 255):
        wv = pix >> 5
        h, w = h0 + (wv // 4) * 32 + (pix & 31), w0 + (wv % 4)
        row = k * H * W + h * W + w
        lst = ent[rp[row]:rp[row + 1]]
        assert len(lst) <= (6 if six[i] else 4)
        for s in range(6):
          if s < len(lst):
            hp, wp = divmod(int(lst[s, 0]), W)
            assert ro[i, k, pix, s] == (wp - cbase) * 81 + (hp - rbase) % H
            assert rw[i, k, pix, s] == lst[s, 1:2].view(np.float32)[0]
          else:
            assert ro[i, k, pix, s] == 0 and rw[i, k, pix, s] == 0.0


def test_no_hip_memset_in_the_library():
  """hipMemsetAsync captured into a hipGraph takes effect on the first launch of the graph only on this ROCm stack (round 4,
  tools/experiments/graph_memset_probe.py): every fill in the library is a kernel (mode::fill_words)."""
  import re
  csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'mode-2022_amd', 'csrc')
  hits = []
  for f in sorted(os.listdir(csrc)):
    if f.endswith(('.hip', '.h')):
      for n, line in enumerate(open(os.path.join(csrc, f)), 1):
        code = line.split('//')[0]
        if re.search(r'hipMemset|hipMemcpy', code):
          hits.append('%s:%d' % (f, n))
  assert not hits, hits
