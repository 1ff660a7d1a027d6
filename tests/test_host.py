"""CPU: host-side mirror of the reference interface (mode-2022_amd/models) -- contract, tables, wiring.

The product has no CPU path.  To check the *wiring* of the module graph without a GPU, the two HIP entry
points it calls are rebound to the oracle inside these tests only (the same trick the golden generator uses
on the reference, SURVEY.md section 8c shim 3)."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

import plain_ops
import recipe
from oracle import mode_ref, sphere_conv_ref

import models
from models.basic import SphereConv
from models.basic.spherical_conv import sphere_conv as sc_mod
from models import mode_disparity as md_mod

GOLDEN = os.path.dirname(os.path.abspath(recipe.__file__))


@pytest.fixture
def oracle_ops(monkeypatch):
  # (SphereConv.forward passes one argument more than the reference op has: `training`, which only picks the arithmetic of the HIP kernels)
  monkeypatch.setattr(sc_mod, 'sphere_conv', lambda *a: sphere_conv_ref.sphere_conv(*a[:8]))
  monkeypatch.setattr(md_mod.HF, 'cost_volume', mode_ref.cost_volume)
  monkeypatch.setattr(md_mod.stage3d, 'conv3', lambda conv, x: conv(x))  # torch CPU conv = what oracle/mode_ref.py uses
  monkeypatch.setattr(md_mod.stage3d, 'head', plain_ops.head)
  monkeypatch.setattr(md_mod.stage3d, 'bn_act', md_mod.stage3d.bn_act_torch)


@pytest.fixture(scope='module')
def tiny_model():
  return models.ModeDisparity(16, 'Sphere', 64, 32, 'Cassini')


# ------------------------------------------------------------------ contract
def test_state_dict_matches_reference_manifest(tiny_model):
  manifest = recipe.load_manifest()
  sd = tiny_model.state_dict()
  assert len(manifest) == 483
  assert [(k, tuple(v.shape)) for k, v in sd.items()] == manifest
  assert not any('position' in k for k in sd)
  n_params = sum(p.numel() for p in tiny_model.parameters())
  assert n_params == 5489280


def test_constructor_contract():
  with pytest.raises(NotImplementedError):
    models.ModeDisparity(16, conv='Nope')
  m = models.ModeDisparity(192, 'Regular')
  assert any(k.startswith('feature_extraction.branch4') for k in m.state_dict())
  assert callable(models.initModelPara) and callable(models.loadStackHourglassOnly)
  models.initModelPara(m, 'default')
  with pytest.raises(AssertionError):
    SphereConv(10, 30, 'ERP', 1, 1, 3)  # width must be 2*height
  with pytest.raises(AssertionError):
    SphereConv(8, 16, 'Fisheye', 1, 1, 3)
  s = SphereConv(16, 8, 'Cassini', 4, 6, 3, 1, 1)
  assert (s.in_height, s.in_width) == (8, 16) and s.bias is None and s.kernel_size == (3, 3)
  assert s.getPosition().shape == (1, 18, 16, 8) and 'position' not in s.state_dict()
  assert abs(float(s.weight.abs().max())) <= 1 / np.sqrt(4 * 9) + 1e-7


def test_psmnet_init_statistics(tiny_model):
  w = tiny_model.dres0[0][0].weight
  assert abs(float(w.std()) - np.sqrt(2.0 / (27 * 32))) < 2e-3
  bn = tiny_model.dres0[0][1]
  assert float(bn.weight.min()) == 1.0 and float(bn.bias.abs().max()) == 0.0


def test_no_cpu_path(tiny_model):
  x = torch.zeros(1, 3, 64, 32)
  with pytest.raises(NotImplementedError, match='Only support cuda tensor'):
    tiny_model(x, x)
  with pytest.raises(ValueError):
    sc_mod.SphereConvFunction.apply(torch.zeros(3, 16, 8), None, torch.zeros(1, 3, 3, 3))
  with pytest.raises(NotImplementedError):
    md_mod.HF.cost_volume(torch.zeros(1, 2, 4, 4), torch.zeros(1, 2, 4, 4), 2)


def test_load_stack_hourglass_only(tmp_path, tiny_model):
  other = {k: torch.full_like(v, 0.5) for k, v in tiny_model.state_dict().items()}
  other['not.in.model'] = torch.zeros(1)
  path = str(tmp_path / 'ckpt.tar')
  torch.save({'state_dict': other}, path)
  m = models.ModeDisparity(16, 'Sphere', 64, 32, 'Cassini')
  before = m.feature_extraction.firstconv[0][0].weight.clone()
  models.loadStackHourglassOnly(m, path)
  assert torch.equal(m.feature_extraction.firstconv[0][0].weight, before)
  assert float(m.dres2.conv5[0].weight.min()) == 0.5 and float(m.classif3[2].weight.max()) == 0.5


# ------------------------------------------------------------------ sampling table (a5)
@pytest.mark.parametrize('typ,ih,iw', [('ERP', 8, 16), ('Cassini', 16, 8)])
def test_product_table_small(golden, typ, ih, iw):
  g = golden('positions.npz')['%s_%dx%d' % (typ, ih, iw)]
  p = SphereConv(ih, iw, typ, 1, 1, 3, 1, 1).position.numpy()
  assert p.dtype == np.float32 and np.array_equal(p, g)


@pytest.mark.parametrize('key', ['Cassini_256x128', 'Cassini_128x64', 'ERP_128x256', 'Cassini_512x256'])
def test_product_table_sha(key):
  with open(os.path.join(GOLDEN, 'positions_meta.json')) as f:
    meta = json.load(f)[key]
  typ, dims = key.split('_')
  ih, iw = map(int, dims.split('x'))
  p = SphereConv(ih, iw, typ, 1, 1, 3, 1, 1).position.numpy()
  assert list(p.shape) == meta['shape']
  assert hashlib.sha256(np.ascontiguousarray(p).tobytes()).hexdigest() == meta['sha256']


def test_table_is_shared_between_layers(tiny_model):
  convs = [m for m in tiny_model.modules() if isinstance(m, SphereConv)]
  assert len(convs) == 16
  assert all(c.position is convs[0].position for c in convs)


# ------------------------------------------------------------------ wiring of the module graph
def _load_bn(sd, z):
  for k in z.files:
    if k.startswith('bn/'):
      sd[k[3:]] = torch.from_numpy(z[k]).clone()


def test_wiring_train_matches_reference(golden, oracle_ops, tiny_model):
  z = golden('model_tiny.npz')
  maxdisp, H, W, B, seed = [int(v) for v in z['cfg']]
  tiny_model.load_state_dict(recipe.recipe_state(recipe.load_manifest(), seed))
  left, right = recipe.recipe_images(B, H, W, seed + 1)
  gt = recipe.recipe_disparity(B, H, W, seed + 2, maxdisp)
  mask = ~torch.isnan(gt)
  tiny_model.train()
  tiny_model.zero_grad()
  preds = tiny_model(left, right)
  for i, p in enumerate(preds):
    assert p.shape == (B, 1, H, W)
    assert np.abs(p.detach().numpy() - z['train/pred%d' % (i + 1)]).max() < 1e-3
  loss = mode_ref.training_loss(preds, gt, mask)
  assert abs(float(loss.detach()) - float(z['train/loss'])) < 1e-4 * float(z['train/loss'])
  loss.backward()
  grads = dict(tiny_model.named_parameters())
  for n, s in zip(z['train/grad_names'], z['train/grad_abs_sum']):
    g = grads[str(n)].grad
    assert abs(float(g.double().abs().sum()) - s) <= 2e-3 * s + 1e-7, n
  # running statistics were updated with momentum 0.1 exactly once
  assert int(tiny_model.dres0[0][1].num_batches_tracked) == 1


def test_wiring_eval_and_confidence(golden, oracle_ops, tiny_model):
  z = golden('model_tiny.npz')
  maxdisp, H, W, B, seed = [int(v) for v in z['cfg']]
  sd = recipe.recipe_state(recipe.load_manifest(), seed)
  _load_bn(sd, z)
  tiny_model.load_state_dict(sd)
  left, right = recipe.recipe_images(B, H, W, seed + 1)
  tiny_model.eval()
  with torch.no_grad():
    pred = tiny_model(left, right)
    tiny_model.out_conf = True
    pred2, conf = tiny_model(left, right)
    tiny_model.out_conf = False
  assert torch.equal(pred, pred2) and conf.shape == pred.shape
  assert np.abs(pred.numpy() - z['eval/pred3']).max() < 1e-3
  assert np.abs(conf.numpy() - z['eval/conf']).max() < 1e-4


def test_module_prefix_checkpoints_load(tiny_model):
  """Checkpoints saved under nn.DataParallel carry a 'module.' prefix (train_disparity.py:91-94)."""
  wrapped = torch.nn.DataParallel(tiny_model)
  sd = wrapped.state_dict()
  assert all(k.startswith('module.') for k in sd) and len(sd) == 483
  wrapped.load_state_dict(sd)


def test_bn_groups_is_per_thread_state():
  """nn.DataParallel runs one forward() per GPU on its own Python thread: the statistics grouping of the paired extractor pass
  must not leak between threads (a module global did).  Thread A sits inside bn_groups(2) while thread B enters and
  leaves its own context; each sees only its own value, and both see 1 outside."""
  import threading
  from models import stage3d
  seen = {}
  a_inside, b_done = threading.Event(), threading.Event()

  def thread_a():
    with stage3d.bn_groups(2):
      a_inside.set()
      b_done.wait(10)
      seen['a_inside_after_b_left'] = stage3d.current_bn_groups()
    seen['a_after'] = stage3d.current_bn_groups()

  def thread_b():
    a_inside.wait(10)
    seen['b_before'] = stage3d.current_bn_groups()
    with stage3d.bn_groups(3):
      seen['b_inside'] = stage3d.current_bn_groups()
    seen['b_after'] = stage3d.current_bn_groups()
    b_done.set()

  ts = [threading.Thread(target=thread_a), threading.Thread(target=thread_b)]
  for t in ts:
    t.start()
  for t in ts:
    t.join()
  assert seen == dict(a_inside_after_b_left=2, a_after=1, b_before=1, b_inside=3, b_after=1)
  assert stage3d.current_bn_groups() == 1
  assert not hasattr(stage3d, '_bn_groups')  # no module-level mutable left


def test_hourglass_signature_is_the_references():
  import inspect
  sig = inspect.signature(md_mod.hourglass.forward)
  pos = [p.name for p in sig.parameters.values() if p.kind == p.POSITIONAL_OR_KEYWORD]
  assert pos == ['self', 'x', 'presqu', 'postsqu']
  assert sig.parameters['residual'].kind == inspect.Parameter.KEYWORD_ONLY


def test_product_reads_no_environment_switches():
  """Backend / restructuring switches are not configuration: nothing under mode-2022_amd/models or mode_hip/functional.py
  reads os.environ (VERDICT r1 design smell)."""
  root = os.path.join(os.path.dirname(GOLDEN), '..', 'mode-2022_amd')
  for sub in ('models', os.path.join('mode_hip', 'functional.py')):
    path = os.path.join(root, sub)
    files = [path] if path.endswith('.py') else [os.path.join(d, f) for d, _, fs in os.walk(path) for f in fs if f.endswith('.py')]
    for f in files:
      assert 'environ' not in open(f).read().replace('reads no environment', '').replace('read no environment', ''), f


def test_seeded_initialisation_is_the_references(golden):
  """Under the same torch.manual_seed the drop-in modules construct bit-identical weights to the imported reference (fixture:
  tests/golden/seeded_init.json from make_golden_seeded_init.py): same RNG draws in the same order -- including the 1x1
  `downsample` projections ModeFusion's reference builds and never registers."""
  import sys
  sys.path.insert(0, GOLDEN)
  from make_golden_seeded_init import CASES, digest
  with open(os.path.join(GOLDEN, 'seeded_init.json')) as f:
    want = json.load(f)
  for tag, (cls, args) in CASES.items():
    torch.manual_seed(want['seed'])
    assert digest(getattr(models, cls)(*args).state_dict()) == want[tag], tag


def test_conv_arithmetic_switch_and_split_predicate():
  """functional.CONV_ARITH is a plain process-wide setting with two values; the library's predicate for the split kernels needs no
  GPU (stride 1, <= 32 output channels of the GEMM, reduction channels a multiple of 8)."""
  import mode_hip
  from mode_hip import functional as HF
  assert HF.CONV_ARITH == 'bf16x6'
  with pytest.raises(ValueError):
    HF.set_conv_arith('bf16')
  HF.set_conv_arith('f32')
  assert HF.CONV_ARITH == 'f32'
  HF.set_conv_arith('bf16x6')
  lib = mode_hip.lib()
  assert lib.mode_conv3d_split_supported(32, 32, 1, 0) == 1 and lib.mode_conv3d_split_supported(32, 32, 1, 1) == 1
  assert lib.mode_conv3d_split_supported(64, 32, 1, 0) == 1 and lib.mode_conv3d_split_supported(64, 32, 1, 1) == 1
  assert lib.mode_conv3d_split_supported(32, 96, 1, 0) == 0 and lib.mode_conv3d_split_supported(12, 32, 1, 0) == 0
  assert lib.mode_conv3d_split_supported(32, 32, 2, 0) == 0 and lib.mode_conv3d_split_supported(32, 1, 1, 0) == 0
  assert lib.mode_conv3d_wpack_bytes(32, 32) >= 4 * 14 * 3 * 64 * 16


def test_bench_labels_on_the_split_path():
  """bench.py prices a dominant kernel against the bf16 pipe (2 500 / 6 TFLOP/s) only when its layer runs on the split kernels."""
  import importlib.util
  spec = importlib.util.spec_from_file_location('bench_module', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py'))
  bench = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(bench)
  assert bench._on_split_path('conv3d_fwd[32->32 s1 48x256x128]')
  assert bench._on_split_path('conv3d_bwd_data[64->64 s1 24x128x64]')
  assert bench._on_split_path('conv3d_fwd[32->64 s2 48x256x128]') and bench._on_split_path('conv3d_bwd_data[32->64 s2 48x256x128]')  # round 3
  assert bench._on_split_path('conv3d_bwd_weight[32->64 s2 48x256x128]') and bench._on_split_path('conv3d_bn_eval[32->64 s2 48x256x128]')
  assert not bench._on_split_path('conv3d_bwd_weight[32->32 s2 48x256x128]')  # gy in blocks of 64 channels
  assert bench._on_split_path('deconv3d_fwd') and not bench._on_split_path('deconv3d_bn_eval')
  assert not bench._on_split_path('conv3d_fwd[32->1 s1 48x256x128]')
  assert bench._on_split_path('conv3d_bwd_weight[32->32 s1 64x512x256]') and bench._on_split_path('conv2d_fwd[64->64 d2 256x128]')
  assert bench._on_split_path('conv2d_bwd_weight[20->40 d1 26x70]') and not bench._on_split_path('conv2d_fwd[20->40 d1 26x70]')
  # the spherical layers of the extractor (64 / 128 -> 128): compact-window tiles on the split kernels; the integer-table layers that share
  # the labels (32 -> 288 tap products, the 64 -> 64 stride-2 layer) stay on the fp32 gather kernels
  assert bench._on_split_path('sphere_conv_bwd_weight[128->128 256x128]') and bench._on_split_path('sphere_conv_bwd_data[64->128 256x128]')
  assert bench._on_split_path('sphere_conv_fwd[128->128 256x128]') and not bench._on_split_path('sphere_conv_fwd[32->288 256x128]')
  assert not bench._on_split_path('sphere_conv_fwd[64->64 512x256]')
  assert bench.kernel_of('conv3d_bwd_data[32->64 s2 48x256x128]', 'bf16x6') == 'deconv3d_split_kernel'
  assert bench.kernel_of('conv3d_fwd[32->64 s2 48x256x128]', 'bf16x6') == 'conv3d_s2_split_kernel'
  assert bench.kernel_of('conv3d_fwd[32->64 s2 48x256x128]', 'f32') == 'conv3d_kernel'
  assert bench.kernel_of('conv3d_bwd_weight[64->64 s2 24x128x64]', 'bf16x6') == 'conv3d_bww_s2_split_kernel'
  assert abs(bench.MFMA_BF16_PEAK_TFLOPS / 6.0 - 416.67) < 0.01


def test_erp_tables_are_planned_through_their_transpose():
  """sphereType='ERP' on the windowed kernels (host logic only): an ERP table is not plannable as it stands (its shift-invariant
  axis is W), its transpose is the Cassini table bit for bit (sphere_conv.py:226-236) and gets the Cassini plan -- so the ERP
  operator runs the `_t` kernels on the NCHW tensors themselves (mode_hip.functional.sphere_native_t)."""
  from mode_hip import functional as HF
  from oracle import mode_ref
  for h, w in ((128, 256), (256, 512)):  # the quarter-resolution maps of 512 x 1024 and 1024 x 2048 ERP pairs
    erp = mode_ref.sphere_position(h, w, 'ERP').contiguous()
    cas = mode_ref.sphere_position(h, w, 'Cassini').contiguous()
    assert HF.sphere_plan(erp, 3, 3) is None
    post = HF.sphere_native_t(erp, 3, 3)
    assert post is not None and torch.equal(post, cas)
    assert HF.sphere_plan(post, 3, 3)[1] == HF.sphere_plan(cas, 3, 3)[1]
    assert HF.sphere_native_t(cas, 3, 3) is None and HF.sphere_uses_transposed_copies(cas, 3, 3) and not HF.sphere_uses_transposed_copies(erp, 3, 3)


def test_grad_carrier_checks_that_both_consumers_ran():
  """functional.GradCarrier on a toy pair of autograd functions (CPU): a full backward adds the first consumer's gradient inside the
  second; a backward that reaches only one consumer raises at its end and leaves nothing parked; a gradient left over from an aborted
  pass is recognised by its owner and dropped."""
  from mode_hip import functional as HF

  class Twice(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, c):
      ctx.c = c
      return x * 2

    @staticmethod
    def backward(ctx, g):
      prev = ctx.c.take(ctx)
      gx = g * 2
      if prev is not None:
        return gx + prev, None
      if ctx.c.leave(gx, ctx):
        return None, None
      return gx, None

  x = torch.ones(3, requires_grad=True)
  c = HF.GradCarrier()
  c.arm(True)
  c.arm(True)
  h = x * 1.0
  y1, y2 = Twice.apply(h, c), Twice.apply(h, c)
  (y1.sum() + y2.sum()).backward(retain_graph=True)
  assert torch.equal(x.grad, torch.full((3,), 4.0)) and c.grad is None
  x.grad = None
  with pytest.raises(RuntimeError, match='GradCarrier'):
    y1.sum().backward(retain_graph=True)
  assert c.grad is None
  x.grad = None
  (y1.sum() + y2.sum()).backward(retain_graph=True)
  assert torch.equal(x.grad, torch.full((3,), 4.0))
  # a leftover whose owner asks first again (the pass that parked it died before its callbacks ran): dropped, not added
  owner = object()
  c.grad, c.owner = torch.full((3,), 100.0), owner
  assert c.take(owner) is None and c.grad is None
  c.grad, c.owner = torch.full((3,), 100.0), owner
  assert c.take(object()) is not None  # the partner would take it
  assert c.leave(torch.ones(3), owner) is False and c.grad is None  # outside a backward pass nothing is parked


def test_build_notices_changed_compile_flags(tmp_path, monkeypatch):
  """objects compiled with other flags (a MODE_HIP_DEFINES debug build, a per-file flag) must not be taken for up to
  date, and a compile that fails after the flags changed must not leave a stamp that says they match: the flags hash is stored per
  object and written only after that object compiled."""
  import stat
  from mode_hip import build as hb
  src = tmp_path / 'k.hip'
  src.write_text('// nothing')
  fake = tmp_path / 'fakecc'
  fake.write_text('#!/bin/sh\nfor a in "$@"; do [ "$a" = "-DFAIL" ] && exit 1; prev2="$prev"; prev="$a"; done\n: > "$prev"\n')
  fake.chmod(fake.stat().st_mode | stat.S_IXUSR)
  obj_dir = tmp_path / 'obj'
  obj_dir.mkdir()
  monkeypatch.setattr(hb, 'OBJ', str(obj_dir))
  monkeypatch.setattr(hb, 'HIPCC', str(fake))
  monkeypatch.setattr(hb, '_deps_mtime', lambda: 0.0)
  obj, compiled = hb._compile(str(src), False)
  assert compiled and os.path.exists(obj) and os.path.exists(obj + '.flags')
  assert hb._compile(str(src), False) == (obj, False)  # up to date
  monkeypatch.setattr(hb, 'FLAGS', hb.FLAGS + ['-DMODE_TAPTIME'])
  assert hb._compile(str(src), False) == (obj, True)  # other flags: rebuilt
  assert hb._compile(str(src), False) == (obj, False)
  monkeypatch.setitem(hb.FILE_FLAGS, 'k.hip', ['-DX'])
  assert hb._compile(str(src), False) == (obj, True)  # a per-file flag counts
  monkeypatch.setitem(hb.FILE_FLAGS, 'k.hip', ['-DFAIL'])
  with pytest.raises(RuntimeError):
    hb._compile(str(src), False)
  monkeypatch.setitem(hb.FILE_FLAGS, 'k.hip', ['-DX', '-DY'])
  assert hb._compile(str(src), False) == (obj, True)  # the failed attempt left no matching stamp behind
  os.remove(obj + '.flags')
  assert hb._compile(str(src), False) == (obj, True)  # an object of unknown origin


def test_isa_loops_reads_a_listing(tmp_path, monkeypatch, capsys):
  """tools/isa_loops.py on a hand-written listing: it finds the loop with the MFMAs, the vmcnt(0) among its waits, the scratch reload, and
  the run of vector instructions in front of the second MFMA."""
  import importlib.util
  spec = importlib.util.spec_from_file_location('isa_loops', os.path.join(os.path.dirname(__file__), '..', 'tools', 'isa_loops.py'))
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  body = ['_Z6kernelPf:                            ; @_Z6kernelPf', '\ts_load_dwordx2 s[0:1], s[4:5], 0x0', '.LBB0_1:                 ; =>This Inner Loop Header: Depth=1',
          '\tglobal_load_dword v1, v0, s[0:1]', '\tglobal_load_dword v2, v0, s[0:1] offset:4', '\ts_waitcnt vmcnt(1)',
          '\tv_mfma_f32_32x32x16_bf16 a[0:15], v[4:7], v[8:11], a[0:15]'] + ['\tv_add_f32_e32 v3, v3, v1'] * 10 + \
         ['\tscratch_load_dword v9, off, off offset:8', '\ts_waitcnt vmcnt(0)', '\tv_mfma_f32_32x32x16_bf16 a[16:31], v[4:7], v[8:11], a[16:31]',
          '\ts_barrier', '\ts_cbranch_scc1 .LBB0_1', '\ts_endpgm', '.Lfunc_end0:']
  (tmp_path / 'k.s').write_text('\n'.join(body) + '\n')
  monkeypatch.setattr(mod, 'ASM', str(tmp_path))
  mod.scan('')
  out = capsys.readouterr().out
  assert 'mfma    2' in out and 'vmcnt(0)  1 of   2 waits' in out and 'loads   3' in out and 'scratch  1' in out
  mod.timeline('kernel')
  assert '0:Lx2 0:w1 1:L 1:w0 2:B' in capsys.readouterr().out
  mod.bursts('kernel')
  assert '(1, (10, 0, 1, 1))' in capsys.readouterr().out
  mod.chains('kernel')
  assert '[(2, 2)]' in capsys.readouterr().out
