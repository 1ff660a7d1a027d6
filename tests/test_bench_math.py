"""CPU: the arithmetic of bench.py's `roofline` / `targets` block (no GPU, no native call: the split-path predicate is injected)."""
import pytest

import bench


@pytest.fixture(autouse=True)
def _three_piece_pricing(request, monkeypatch):
  """The hand-computed numbers below are for the three-piece bf16 arithmetic on every split label (bench.py --no-conv3d-f16); the test
  of the two-piece fp16 pricing of the stride-1 3-D labels opts out."""
  if 'f16_pricing' not in request.keywords:
    monkeypatch.setattr(bench, 'CONV3D_S1_F16', False)
    monkeypatch.setattr(bench, 'SPHERE_FWD_F16', False)
    monkeypatch.setattr(bench, 'CONV2D_F16', False)


def _kern():
  # two labels on the split-bf16 kernels (priced against 2500 / 6 TFLOP/s), one on the fp32 MFMA kernels (157.3), one HBM-bound
  k = {
      'conv3d_fwd[32->32 s1 48x256x128]': dict(calls=12, total_ms=12 * 0.80, flops=12 * 173.95e9, bytes=0),
      'conv3d_bwd_weight[32->32 s1 48x256x128]': dict(calls=6, total_ms=6 * 0.95, flops=6 * 173.95e9, bytes=0),
      'conv3d_fwd[32->64 s2 48x256x128]': dict(calls=6, total_ms=6 * 0.44, flops=6 * 43.5e9, bytes=0),
      'bn_train_fwd[2x32 48x256x128]': dict(calls=10, total_ms=10 * 0.255, flops=0, bytes=10 * 1.208e9),
  }
  for v in k.values():
    v['avg_ms'] = v['total_ms'] / v['calls']
    v['TFLOPs'] = v['flops'] / (v['total_ms'] * 1e9)
    v['GBps'] = v['bytes'] / (v['total_ms'] * 1e6)
  return k


def _split(label):
  return ' s1 ' in label and label.startswith('conv3d')


@pytest.mark.f16_pricing
def test_stride1_labels_are_priced_against_three_mfmas_per_product(monkeypatch):
  """functional.CONV3D_S1_F16 (the default): forward and both gradients of the stride-1 3-D layers run on two fp16 pieces, three MFMAs
  per product -- priced against 2500 / 3, named after the kernel's third template argument; the classifier's single-channel layer,
  the stride-2 layers and the eval label keep their pipes."""
  monkeypatch.setattr(bench, 'CONV3D_S1_F16', True)
  for name in ('conv3d_fwd', 'conv3d_bwd_data', 'conv3d_bwd_weight'):
    assert bench.label_peak(name + '[32->32 s1 48x256x128]', 'bf16x6', _split) == ('mfma', 2500.0 / 3.0, 'TFLOP/s')
  assert bench.label_peak('conv3d_fwd[32->32 s1 48x256x128]', 'f32', _split) == ('mfma', 157.3, 'TFLOP/s')
  # (round 6: the eval label too, unless --no-eval-f16; the stride-2 eval layers stay on three bf16 pieces)
  monkeypatch.setattr(bench, 'CONV3D_EVAL_F16', True)
  assert bench.label_peak('conv3d_bn_eval[32->32 s1 48x256x128]', 'bf16x6', lambda l: True) == ('mfma', 2500.0 / 3.0, 'TFLOP/s')
  assert bench.label_peak('conv3d_bn_eval[32->64 s2 48x256x128]', 'bf16x6', lambda l: True) == ('mfma', 2500.0 / 6.0, 'TFLOP/s')
  monkeypatch.setattr(bench, 'TIMED_BATCH', 1)
  assert bench.kernel_of('conv3d_bn_eval[32->32 s1 48x256x128]', 'bf16x6', lambda l: True) == 'conv3d_split_kernel<1,1,true,16,true>'
  assert bench.kernel_of('conv3d_bn_eval[64->64 s1 12x64x32]', 'bf16x6', lambda l: True) == 'conv3d_split_kernel<1,1,true,8,true>'
  monkeypatch.setattr(bench, 'CONV3D_EVAL_F16', False)
  assert bench.label_peak('conv3d_bn_eval[32->32 s1 48x256x128]', 'bf16x6', lambda l: True) == ('mfma', 2500.0 / 6.0, 'TFLOP/s')
  assert bench.label_peak('conv3d_fwd[32->64 s2 48x256x128]', 'bf16x6', lambda l: True) == ('mfma', 2500.0 / 6.0, 'TFLOP/s')
  monkeypatch.setattr(bench, 'SPHERE_FWD_F16', True)  # (functional.SPHERE_FWD_F16: the windowed spherical forward of the training step too)
  assert bench.label_peak('sphere_conv_fwd[128->128 256x128]', 'bf16x6', lambda l: True) == ('mfma', 2500.0 / 3.0, 'TFLOP/s')
  assert bench.label_peak('sphere_conv_bwd_data[128->128 256x128]', 'bf16x6', lambda l: True) == ('mfma', 2500.0 / 3.0, 'TFLOP/s')
  assert bench.label_peak('sphere_conv_bwd_weight[128->128 256x128]', 'bf16x6', lambda l: True) == ('mfma', 2500.0 / 3.0, 'TFLOP/s')
  assert bench.label_peak('sphere_conv_bn_eval[128->128 256x128]', 'bf16x6', lambda l: True) == ('mfma', 2500.0 / 6.0, 'TFLOP/s')
  monkeypatch.setattr(bench, 'CONV2D_F16', True)  # (functional.CONV2D_F16: the extractor's 3 x 3 layers, forward and input gradient)
  assert bench.label_peak('conv2d_fwd[64->64 d1 256x128]', 'bf16x6', lambda l: True) == ('mfma', 2500.0 / 3.0, 'TFLOP/s')
  assert bench.label_peak('conv2d_bwd_data[128->128 d2 256x128]', 'bf16x6', lambda l: True) == ('mfma', 2500.0 / 3.0, 'TFLOP/s')
  assert bench.label_peak('conv2d_bwd_weight[128->128 d2 256x128]', 'bf16x6', lambda l: True) == ('mfma', 2500.0 / 3.0, 'TFLOP/s')
  monkeypatch.setattr(bench, 'CONV3D_EVAL_F16', False)
  assert bench.label_peak('conv2d_bn_eval[64->64 d1 256x128]', 'bf16x6', lambda l: True) == ('mfma', 2500.0 / 6.0, 'TFLOP/s')
  monkeypatch.setattr(bench, 'CONV3D_EVAL_F16', True)  # (round 6: the eval forward's 3 x 3 layers on two fp16 pieces too)
  assert bench.label_peak('conv2d_bn_eval[64->64 d1 256x128]', 'bf16x6', lambda l: True) == ('mfma', 2500.0 / 3.0, 'TFLOP/s')
  assert bench.label_peak('sphere_conv_fwd[32->288 256x128]', 'bf16x6', lambda l: False)[1] == 157.3  # (the integer-table layers: gather kernels)
  # (round 6: the plain-store fp16 instantiation has a 16-row tile where the volume has >= 4 x 256 of the 8-row tiles -- its own device kernel)
  monkeypatch.setattr(bench, 'TIMED_BATCH', 2)
  monkeypatch.delenv('MODE_SPLIT_TALL', raising=False)
  assert bench.kernel_of('conv3d_fwd[32->32 s1 48x256x128]', 'bf16x6', _split) == 'conv3d_split_kernel<1,0,true,16,false>'
  assert bench.kernel_of('conv3d_bwd_data[64->64 s1 24x128x64]', 'bf16x6', _split) == 'conv3d_split_kernel<1,0,true,8,false>'
  monkeypatch.setattr(bench, 'TIMED_BATCH', 1)
  assert bench.kernel_of('conv3d_fwd[32->32 s1 48x256x128]', 'bf16x6', _split) == 'conv3d_split_kernel<1,0,true,16,false>'
  assert bench.kernel_of('conv3d_fwd[32->32 s1 16x64x128]', 'bf16x6', _split) == 'conv3d_split_kernel<1,0,true,8,false>'
  assert bench.kernel_of('conv3d_fwd[32->32 s1 48x252x128]', 'bf16x6', _split) == 'conv3d_split_kernel<1,0,true,8,false>'
  monkeypatch.setattr(bench, 'CONV3D_EVAL_F16', False)
  assert bench.kernel_of('conv3d_bn_eval[32->32 s1 48x256x128]', 'bf16x6', lambda l: True) == 'conv3d_split_kernel<1,1,false,8,false>'
  assert bench.kernel_of('conv3d_bwd_weight[32->32 s1 48x256x128]', 'bf16x6', _split) == 'conv3d_bww_split_kernel'
  k = _kern()
  f = bench.blended_mfma_fraction(k, 'bf16x6', _split)
  need = (12 + 6) * 173.95e9 / (2500e12 / 3) + 6 * 43.5e9 / 157.3e12
  took = (12 * 0.80 + 6 * 0.95 + 6 * 0.44) * 1e-3
  assert abs(f - need / took) < 1e-12


def test_each_label_is_priced_against_the_pipe_it_runs_on():
  assert bench.label_peak('conv3d_fwd[32->32 s1 48x256x128]', 'bf16x6', _split) == ('mfma', 2500.0 / 6.0, 'TFLOP/s')
  assert bench.label_peak('conv3d_fwd[32->32 s1 48x256x128]', 'f32', _split) == ('mfma', 157.3, 'TFLOP/s')
  assert bench.label_peak('conv3d_fwd[32->64 s2 48x256x128]', 'bf16x6', _split) == ('mfma', 157.3, 'TFLOP/s')
  assert bench.label_peak('bn_train_fwd[2x32 48x256x128]', 'bf16x6', _split) == ('hbm', 8000.0, 'GB/s')


def test_blended_regulariser_fraction_is_a_fraction():
  k = _kern()
  f = bench.blended_mfma_fraction(k, 'bf16x6', _split)
  # by hand: time the pipes' peaks need / time taken
  need = (12 + 6) * 173.95e9 / (2500e12 / 6) + 6 * 43.5e9 / 157.3e12
  took = (12 * 0.80 + 6 * 0.95 + 6 * 0.44) * 1e-3
  assert abs(f - need / took) < 1e-12 and 0.0 < f < 1.0
  # round 2's formula (all flops against the fp32 peak) exceeds 1 on the same numbers -- the defect this replaces
  old = sum(v['flops'] for n, v in k.items() if n.startswith('conv3d')) / (took * 1e12) / 157.3
  assert old > 1.0
  # in f32 mode every label is priced against the fp32 MFMA peak
  f32 = bench.blended_mfma_fraction(k, 'f32', _split)
  assert abs(f32 - old) < 1e-12


def test_by_kernel_groups_labels_of_one_device_kernel():
  k = _kern()
  k['conv3d_bwd_data[32->32 s1 48x256x128]'] = dict(k['conv3d_fwd[32->32 s1 48x256x128]'])
  g = bench.by_kernel(k, 'bf16x6', _split)
  a = g['conv3d_split_kernel<1,0,false,8,false>']
  assert a['calls'] == 24 and abs(a['total_ms'] - 2 * 12 * 0.80) < 1e-9 and len(a['labels']) == 2
  assert abs(a['need_s'] / (a['total_ms'] * 1e-3) - (173.95e9 / (2500e12 / 6)) / 0.80e-3) < 1e-9
  assert g['conv3d_bww_split_kernel']['calls'] == 6 and g['conv3d_kernel']['calls'] == 6
  assert g['bn_stats_kernel+bn_apply_kernel']['bound'] == 'hbm'
  hb = g['bn_stats_kernel+bn_apply_kernel']
  assert abs(hb['need_s'] - 10 * 1.208e9 / 8000e9) < 1e-12


def test_roofline_block_names_the_dominant_device_kernel():
  """`roofline` = the device kernel with the largest total time over all its layer shapes (rocprofv3 --stats' first row), its
  fraction from ALL its launches; `by_label` = the single heaviest (operator, shape) label."""
  k = _kern()
  k['conv3d_bwd_data[32->32 s1 48x256x128]'] = dict(k['conv3d_fwd[32->32 s1 48x256x128]'])
  k['conv3d_fwd[64->64 s1 24x128x64]'] = dict(calls=6, total_ms=6 * 0.40, flops=6 * 86.97e9, bytes=0, avg_ms=0.40, TFLOPs=86.97 / 0.40, GBps=0.0)
  for v in k.values():
    v['bytes_per_call'] = v['bytes'] / v['calls']
    v['flops_per_call'] = v['flops'] / v['calls']
  r = bench.roofline_block(k, 'bf16x6', 2, 2, 'test', _split)
  assert r['kernel'] == 'conv3d_split_kernel<1,0,false,8,false>' and r['bound'] == 'mfma' and r['unit'] == 'TFLOP/s'
  # launches, as rocprofv3 counts them (a 64-channel stride-1 layer is one launch with two y-slices)
  assert r['calls'] == 30 and abs(r['avg_ms'] - (24 * 0.80 + 6 * 0.40) / 30) < 1e-12 and abs(r['ms_per_step'] - (24 * 0.80 + 6 * 0.40) / 2) < 1e-12
  flops, sec = 24 * 173.95e9 + 6 * 86.97e9, (24 * 0.80 + 6 * 0.40) * 1e-3
  assert abs(r['achieved'] - flops / sec / 1e12) < 1e-9 and abs(r['peak'] - 2500.0 / 6.0) < 1e-12
  assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12 and 0 < r['frac'] < 1
  assert abs(r['algorithmic_per_launch'] - flops / 30) < 1e-3
  assert r['traffic_label'] in ('conv3d_fwd[32->32 s1 48x256x128]', 'conv3d_bwd_data[32->32 s1 48x256x128]')
  assert r['traffic'] is not None and r['traffic'] > 805306368, 'the calibrated PMC figure of the heaviest shape (profiles/traffic.json)'
  b = r['by_label']
  assert b['kernel'] in ('conv3d_fwd[32->32 s1 48x256x128]', 'conv3d_bwd_data[32->32 s1 48x256x128]') and abs(b['frac'] - (173.95 / 0.80) / (2500.0 / 6.0)) < 1e-9


def test_kernel_of_names_device_kernels_that_really_ran():
  """kernel_of() must name the device kernel a label's launches run on -- every name it returns for the labels of a recorded
  bench line is a kernel of the rocprofv3 --kernel-trace --stats file recorded with the same build (the newest profiles/r05*_ pair), in both the
  split and the fp32 pricing of the label (the real predicate, mode_hip's host-side *_supported queries: no GPU needed)."""
  import csv
  import json
  import os
  prof = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles')
  # the newest recorded pair (bench line, kernel stats of the same build): profiles/<tag>_bench.json + <tag>_rocprofv3_kernel_stats_*.csv
  tags = sorted(f[:-len('_bench.json')] for f in os.listdir(prof) if f.endswith('_bench.json') and
                os.path.exists(os.path.join(prof, f[:-len('_bench.json')] + '_rocprofv3_kernel_stats_bench_graph_steps2.csv')))
  tag = tags[-1]
  assert tag >= 'r05', tag
  with open(os.path.join(prof, tag + '_bench.json')) as f:
    labels = list(json.load(f)['kernels'])
  with open(os.path.join(prof, tag + '_rocprofv3_kernel_stats_bench_graph_steps2.csv')) as f:
    ran = [row['Name'] for row in csv.DictReader(f)]
  assert len(labels) > 60 and len(ran) > 60
  missing = []
  for label in labels:
    if label.startswith('cost_volume_fwd'):
      continue  # timed standalone after the step (the model folds the volume away): not in the step's trace
    for part in bench.kernel_of(label, 'bf16x6').split('+'):
      base = part.split('<')[0]
      if not any(('::' + base + '(') in n or ('::' + base + '<') in n or n.startswith(base + '(') for n in ran):
        missing.append((label, part))
  assert not missing, missing
  # the split gradients of the gnomonic layers are their own kernels; the integer-table layers under the same labels are not split
  assert bench.kernel_of('sphere_conv_bwd_data[128->128 256x128]', 'bf16x6') == 'sphere_bwd_data_split_kernel'
  assert bench.kernel_of('sphere_conv_bwd_weight[128->128 256x128]', 'bf16x6') == 'sphere_bww_split_kernel'
  assert bench.kernel_of('sphere_conv_fwd[32->288 256x128]', 'bf16x6') == 'sphere_fwd_kernel'
  assert bench.kernel_of('sphere_conv_bwd_data[128->128 256x128]', 'f32') == 'sphere_bwd_data_adj9_kernel'


def test_clock_sampling_parses_rocm_smi_and_never_raises(monkeypatch):
  """bench.sample_clocks: the engine clock / socket power of THIS device out of `rocm-smi --showclocks --showpower` (the format of the
  MI355X box, profiles/r06v_clocks_under_load.txt), polled while `step` runs; None when the tool prints nothing for the device."""
  import subprocess
  import types
  text = ('GPU[0]\t\t: fclk clock level: 0: (1250Mhz)\nGPU[0]\t\t: mclk clock level: 0: (2000Mhz)\nGPU[0]\t\t: sclk clock level: S: (2165Mhz)\n'
          'GPU[1]\t\t: sclk clock level: S: (95Mhz)\n====== Power Consumption ======\nGPU[0]\t\t: Current Socket Graphics Package Power (W): 1208.0\n'
          'GPU[1]\t\t: Current Socket Graphics Package Power (W): 239.0\n')
  monkeypatch.setattr(bench.os.path, 'exists', lambda p: True)
  monkeypatch.setattr(subprocess, 'run', lambda *a, **k: types.SimpleNamespace(stdout=text, returncode=0))
  monkeypatch.setattr(bench.time, 'sleep', lambda s: None)
  steps = []
  c = bench.sample_clocks(lambda: steps.append(1), lambda: None, 0, seconds=0.05)
  assert c['sclk_mhz_under_step'] == 2165 and c['socket_power_w_under_step'] == 1208.0 and c['sclk_mhz_idle_after'] == 2165 and len(steps) == c['extra_steps']
  assert bench.sample_clocks(lambda: None, lambda: None, 1, seconds=0.05)['sclk_mhz_under_step'] == 95
  assert bench.sample_clocks(lambda: None, lambda: None, 3, seconds=0.05) is None  # no such device in the listing
