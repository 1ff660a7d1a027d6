#!/usr/bin/env python3
"""bench.py -- MODE disparity stage on MI355X: stereo pairs/s, fwd+bwd(+Adam), synthetic Cassini 1024x512 (= 512x1024 ERP),
192 disparities, batch 2 per GPU (BASELINE.json configs[2]; configs[3] when launched on 8 GPUs).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Rank 0 prints ONE JSON line.  A "step" is one training iteration of ModeDisparity over one synthetic batch that is already
resident in HBM: zero-grad, forward (3 heads), masked smooth-L1 loss 0.5/0.7/1.0, backward, gradient all-reduce (N > 1),
Adam step.  By default zero-grad + forward + loss + backward are replayed as one hipGraph (--launch graph).  `roofline`
describes the dominant hand-written kernel, timed with HIP events on its launch stream (inside the timed region with
--launch eager; over --profile-steps eager steps right after it with --launch graph, where events cannot sit inside the
replayed graph); `cpu_baseline` times the CPU oracle (a port, not the product) on this host's cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'mode-2022_amd'), os.path.join(ROOT, 'tests', 'golden')):
  if p not in sys.path:
    sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.nn.functional as F  # noqa: E402

# peaks from /opt/skills/guides/MI355X_MICROARCH.md (chip-level parameters)
HBM_PEAK_GBPS = 8000.0
MFMA_F32_PEAK_TFLOPS = 157.3
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense (bf16 and fp16 alike); a split-bf16 fp32 product costs six bf16 MFMAs (csrc/conv3d_split.hip)
# The stride-1 3-D layers in training run on TWO fp16 pieces and three MFMAs per product (functional.CONV3D_S1_F16): their
# labels are priced against the dense peak / 3.  Set from --no-conv3d-f16 in main().
CONV3D_S1_F16 = True
# ... and so do the windowed spherical forward and both of its gradients in the training step (functional.SPHERE_FWD_F16 / SPHERE_BWD_F16).
# Set from --no-sphere-f16.
SPHERE_FWD_F16 = True
CONV2D_F16 = True  # (functional.CONV2D_F16: forward and input gradient of the extractor's stride-1 3 x 3 layers; --no-conv2d-f16)
CONV3D_EVAL_F16 = True  # (functional.CONV3D_EVAL_F16: the stride-1 3-D layers of an INFERENCE forward on the same arithmetic; --no-eval-f16)


def _on_f16_path(label):
  import re
  if CONV3D_S1_F16 and re.match(r'(conv3d_fwd|conv3d_bwd_data|conv3d_bwd_weight)\[(\d+)->(\d+) s1 ', label) and not re.search(r'->1 ', label):
    return True
  if CONV3D_EVAL_F16 and re.match(r'conv3d_bn_eval\[(\d+)->(\d+) s1 ', label):
    return True
  if CONV3D_EVAL_F16 and re.match(r'conv2d_bn_eval\[', label):  # (functional.CONV2D_EVAL_F16: the same switch in this script)
    return True
  m = re.match(r'sphere_conv_(fwd|bwd_data|bwd_weight)\[(\d+)->(\d+) ', label)  # (the windowed 3x3 gnomonic layers: kernel_of below)
  if SPHERE_FWD_F16 and m and int(m.group(3)) % 128 == 0 and int(m.group(2)) % 16 == 0:
    return True
  return bool(CONV2D_F16 and re.match(r'conv2d_(fwd|bwd_data|bwd_weight)\[', label))  # (where they are on the split path at all: label_peak asks that first)
TIMED_BATCH = 2  # (set by main: the layer labels carry no batch size)


def _tall_tile(label):
  """csrc/conv3d_split.hip, conv3d_s1_split: the plain-store fp16 instantiation runs a 2 x 16 x 32 output tile where H % 16 == 0 and the
  2 x 8 x 32 tiling has at least four tiles per CU and output block -- another device kernel (<1,0,true,16>) in rocprofv3's table."""
  import re
  m = re.search(r' s1 (\d+)x(\d+)x(\d+)\]', label)
  if not m or os.environ.get('MODE_SPLIT_TALL', '1')[:1] == '0':
    return False
  d, h, w = (int(v) for v in m.groups())
  return h % 16 == 0 and TIMED_BATCH * ((d + 1) // 2) * (h // 8) * ((w + 31) // 32) >= 4 * 256


KERNEL_BOUND = {'maxpool2x2_fwd': 'hbm', 'maxpool2x2_bwd': 'hbm', 'depth_to_space2': 'hbm', 'space_to_depth2': 'hbm', 'conv1x1_sigmoid_fwd': 'hbm',
                'conv1x1_sigmoid_bwd': 'hbm', 'cost_volume_fwd': 'hbm', 'cost_volume_bwd': 'hbm', 'head_fwd': 'hbm', 'head_bwd': 'hbm', 'bn_train_fwd': 'hbm',
                'bn_train_bwd': 'hbm', 'bn_eval_fwd': 'hbm', 'cost_conv_assemble_fwd': 'hbm', 'cost_conv_assemble_bwd': 'hbm',
                'classif_fwd': 'hbm', 'classif_bwd': 'hbm'}


def parse():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=5)
  ap.add_argument('--warmup', type=int, default=2)
  ap.add_argument('--batch', type=int, default=None, help='pairs per GPU (default 2; --mode fusion: frames per call, default 1)')
  ap.add_argument('--height', type=int, default=1024)
  ap.add_argument('--width', type=int, default=512)
  ap.add_argument('--maxdisp', type=int, default=192)
  ap.add_argument('--mode', default='train', choices=['train', 'eval', 'fusion'],
                  help="'fusion': ModeFusion inference forward at --height x --width (the second half of BASELINE configs[4]; models/mode_fusion.py)")
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--cpu-baseline-only', action='store_true', help='(internal) run the CPU oracle timing and print its JSON')
  ap.add_argument('--cpu-baseline-timeout', type=int, default=420)
  ap.add_argument('--no-kernel-timing', action='store_true')
  ap.add_argument('--launch', default='graph', choices=['graph', 'eager'],
                  help="graph: forward+loss+backward replayed as one hipGraph (default); eager: one launch per kernel")
  ap.add_argument('--vendor-autotune', type=int, default=int(os.environ.get('MODE_VENDOR_AUTOTUNE', '0')),
                  help='1: torch.backends.cudnn.benchmark = True (MIOpen times its solvers for the regular 2-D convolutions)')
  ap.add_argument('--profile-steps', type=int, default=2, help='eager steps with per-kernel HIP-event timing (after the timed region)')
  ap.add_argument('--conv-arith', default='bf16x6', choices=['f32', 'bf16x6'],
                  help="stride-1 3x3x3 layers: 'f32' = fp32 MFMA; 'bf16x6' = fp32 operands split into three bf16 pieces, six bf16 MFMAs "
                  'per product, fp32 accumulation (fp32 accuracy; mode_hip/functional.py CONV_ARITH)')
  ap.add_argument('--value-1gpu', type=float, default=None,
                  help='pairs/s of the same workload on ONE GPU, if known: rank 0 adds value / (N * value_1gpu) to the line')
  ap.add_argument('--fused-bn-stats', action='store_true',
                  help='A/B: BatchNorm statistics of the stride-1 3-D layers in the convolution epilogue (functional.CONV3D_BN_STATS; measured +-0)')
  ap.add_argument('--init', default='recipe', choices=['recipe', 'torch'],
                  help="initial weights: 'recipe' = tests/golden/recipe.py state (SURVEY 8c/8d), 'torch' = the constructor's random init under torch.manual_seed(0)")
  ap.add_argument('--no-fused-loss', action='store_true',
                  help='A/B: the loss of train_disparity.py:151-158 as torch ops on the three predictions instead of ModeDisparity.forward_loss')
  ap.add_argument('--no-conv3d-f16', action='store_true',
                  help='A/B: the stride-1 3-D layers of the training step on three bf16 pieces / six MFMAs per product like every other split '
                       'kernel, instead of two fp16 pieces / three MFMAs with a power-of-two scale per operand tensor (functional.CONV3D_S1_F16)')
  ap.add_argument('--no-eval-f16', action='store_true',
                  help='A/B: the stride-1 3-D and 3 x 3 layers of an inference forward on three bf16 pieces (functional.CONV3D_EVAL_F16 = CONV2D_EVAL_F16 = False)')
  ap.add_argument('--no-sphere-f16', action='store_true',
                  help='A/B: the windowed spherical forward and gradients of the training step on three bf16 pieces (functional.SPHERE_FWD_F16 = SPHERE_BWD_F16 = False)')
  ap.add_argument('--no-conv2d-f16', action='store_true',
                  help='A/B: the stride-1 3 x 3 layers of the training step (forward and both gradients) on three bf16 pieces (functional.CONV2D_F16 = False)')
  ap.add_argument('--no-grad-carriers', action='store_true',
                  help="A/B: autograd's own pairwise accumulation for the tensors with two consumers (functional.GRAD_CARRIERS = False)")
  ap.add_argument('--no-fused-classif', action='store_true',
                  help='A/B: the classifier heads as separate BatchNorm / 32->1 convolution operators (functional.CLASSIF_FUSED = False)')
  ap.add_argument('--no-collective-self-test', action='store_true', help='skip the world-size-1 RCCL all-reduce self-test after the timed region')
  ap.add_argument('--no-clock-sample', action='store_true',
                  help='skip the 4 s of extra steps after the timed region during which rocm-smi is polled for the engine clock and socket power')
  ap.add_argument('--no-eval-b1', action='store_true', help='skip the BASELINE configs[1] leg (eval forward, batch 1) after the timed region')
  ap.add_argument('--dist-backend', default='nccl', choices=['nccl', 'gloo'],
                  help="'nccl' is RCCL on ROCm (xGMI inside the node); 'gloo' lets several ranks share ONE GPU in the tests "
                  '(RCCL refuses two ranks on the same device)')
  args = ap.parse_args()
  if args.batch is None:
    args.batch = 1 if args.mode == 'fusion' else 2
  global TIMED_BATCH
  TIMED_BATCH = args.batch
  return args


def launch_ranks(args):
  """`python bench.py --gpus N` without a launcher around it: start N ranks (one process per GPU) with torch.distributed.run as
  a CHILD process and pass its output and exit code through.  Nothing in this parent has touched the GPU (no HIP call, not even
  torch.cuda.is_available()), and the parent does not exec: it waits for the child."""
  import socket
  import subprocess
  with socket.socket() as sk:
    sk.bind(('127.0.0.1', 0))
    port = sk.getsockname()[1]
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
         '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
  # HSA_ENABLE_IPC_MODE_LEGACY=0: the hosts of this pool only support dmabuf IPC; without it RCCL's intra-node transport (and any
  # device-tensor sharing between the ranks) fails in hipIpcGetMemHandle with "invalid argument".  The image exports it already;
  # it is repeated here so that a launcher with a scrubbed environment still gets it (an explicit setting of the caller wins).
  env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
  return subprocess.call(cmd, env=env)


def _on_split_path(label):
  """Does the kernel behind a profiling label like conv3d_fwd[32->32 s1 48x256x128] or conv2d_bwd_weight[64->64 d1 256x128] run on
  the split-bf16 kernels in bf16x6 mode?"""
  import re
  import mode_hip
  lib = mode_hip.lib()
  m = re.match(r'(conv3d_fwd|conv3d_bwd_data|conv3d_bwd_weight|conv3d_bn_eval)\[(\d+)->(\d+) s(\d) ', label)
  if m:
    which = {'conv3d_bwd_data': 1, 'conv3d_bwd_weight': 2}.get(m.group(1), 0)
    return lib.mode_conv3d_split_supported(int(m.group(2)), int(m.group(3)), int(m.group(4)), which) == 1
  if label.startswith('deconv3d_fwd'):
    return True  # (64 -> 64 and 64 -> 32 in the network: mode_deconv3d_split_supported)
  m = re.match(r'(sphere_conv_fwd|sphere_conv_bwd_data|sphere_conv_bwd_weight|sphere_conv_bn_eval)\[(\d+)->(\d+) (\d+)x(\d+)\]', label)
  if m:
    # the spherical layers of the extractor (3x3 taps on the gnomonic table): compact-window tiles on the split-bf16 kernels, the
    # tiles next to the poles on fp32 MFMA inside the same operator -- priced against the faster pipe (the lower fraction).  The
    # integer-table layers that share these labels (32 -> 288 tap products, the stride-2 layer) stay on the fp32 gather kernels.
    ci, co = int(m.group(2)), int(m.group(3))
    if m.group(1) == 'sphere_conv_bwd_data':
      return co % 128 == 0 and lib.mode_sphere_conv_bwd_data_win_supported(ci, co, 1) == 1
    return co % 128 == 0 and ci % 16 == 0  # (layer4 of the extractor: 64 -> 128 and 128 -> 128)
  m = re.match(r'(conv2d_fwd|conv2d_bwd_data|conv2d_bwd_weight|conv2d_bn_eval)\[(\d+)->(\d+) d(\d) ', label)
  if m:
    if m.group(1) == 'conv2d_bwd_weight':
      return True  # any channel counts
    return lib.mode_conv2d_split_supported(int(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(1) == 'conv2d_bwd_data')) == 1
  return False


def label_peak(label, conv_arith, on_split=None):
  """(bound, peak, unit) a profiling label is priced against: HBM for the bandwidth-bound kernels; for the matrix kernels the fp32
  MFMA peak, or -- where the layer runs on the split-bf16 kernels (`on_split(label)`) -- the dense bf16 peak / 6 (six bf16 MFMAs per
  fp32 product): always the pipe the instructions actually issue on."""
  if KERNEL_BOUND.get(label.split('[')[0], 'mfma') == 'hbm':
    return 'hbm', HBM_PEAK_GBPS, 'GB/s'
  split = conv_arith == 'bf16x6' and (on_split if on_split is not None else _on_split_path)(label)
  if split and _on_f16_path(label):
    return 'mfma', MFMA_BF16_PEAK_TFLOPS / 3.0, 'TFLOP/s'  # two fp16 pieces: three MFMAs per fp32 product
  return 'mfma', (MFMA_BF16_PEAK_TFLOPS / 6.0 if split else MFMA_F32_PEAK_TFLOPS), 'TFLOP/s'


def blended_mfma_fraction(kern, conv_arith, on_split=None, prefixes=('conv3d_', 'deconv3d_')):
  """Time-weighted fraction of the matrix-pipe peak over a group of labels, each priced against ITS pipe:
  sum_k (flops_k / peak_k) / sum_k t_k.  A fraction of the time the pipes could have done the work in -- never above 1 (dividing
  the summed fp32-equivalent flops by the fp32 peak, as round 2 did, exceeds 1 once part of the work runs on the bf16 pipe)."""
  need = total = 0.0
  for k, v in kern.items():
    if not k.startswith(prefixes):
      continue
    peak = label_peak(k, conv_arith, on_split)[1]
    need += v['flops'] / (peak * 1e12)
    total += v['total_ms'] * 1e-3
  return need / total if total > 0 else None


# profiling label -> the device kernel that does its work (names as rocprofv3 --kernel-trace --stats prints them), so that the bench
# line can name the dominant KERNEL next to the dominant (kernel, layer shape) label.
def kernel_of(label, conv_arith, on_split=None):
  import re
  name = label.split('[')[0]
  split = conv_arith == 'bf16x6' and (on_split if on_split is not None else _on_split_path)(label)
  m = re.search(r' s(\d) ', label)
  stride = int(m.group(1)) if m else 1
  if name in ('conv3d_fwd', 'conv3d_bwd_data', 'conv3d_bn_eval'):
    if re.search(r'->1 ', label):
      return 'classif_fwd_kernel' if name != 'conv3d_bwd_data' else 'conv3d_co1_bwd_data_kernel'  # (Ci <= 32: classif_head.hip's forward without the BatchNorm prologue)
    if split and stride == 2:
      return 'deconv3d_split_kernel' if name == 'conv3d_bwd_data' else 'conv3d_s2_split_kernel'
    if split:
      if _on_f16_path(label) and name == 'conv3d_bn_eval':  # (<1,2,true,8,true> with a residual: always the 8-row tile)
        return 'conv3d_split_kernel<1,1,true,16,true>' if _tall_tile(label) else 'conv3d_split_kernel<1,1,true,8,true>'
      if _on_f16_path(label):  # (<1,2,true,8,false> where a gradient is added in the store)
        return 'conv3d_split_kernel<1,0,true,16,false>' if _tall_tile(label) else 'conv3d_split_kernel<1,0,true,8,false>'
      return 'conv3d_split_kernel<1,0,false,8,false>' if name != 'conv3d_bn_eval' else 'conv3d_split_kernel<1,1,false,8,false>'  # (<1,2,false,8,false> with a residual)
    return 'conv3d_kernel' if not (name == 'conv3d_bwd_data' and stride == 2) else 'deconv3d_kernel'
  if name == 'conv3d_bwd_weight':
    if re.search(r'->1 ', label):
      return 'conv3d_co1_bwd_weight_kernel'
    if split:
      return 'conv3d_bww_split_kernel' if stride == 1 else 'conv3d_bww_s2_split_kernel'
    return 'conv3d_bwd_weight_ring_kernel' if stride == 1 else 'conv3d_bwd_weight_s2_kernel'
  if name == 'deconv3d_fwd':
    return 'deconv3d_split_kernel' if conv_arith == 'bf16x6' else 'deconv3d_kernel'
  if name == 'deconv3d_bn_eval':  # (the model's 64 -> 64 / 64 -> 32 layers: whole output tiles, the split kernel's own epilogue instantiation)
    return 'deconv3d_split_kernel' if conv_arith == 'bf16x6' else 'deconv3d_kernel'
  if name in ('conv2d_fwd', 'conv2d_bwd_data', 'conv2d_bn_eval'):
    return 'conv2d_split_kernel' if split else 'conv2d_kernel'
  if name == 'conv2d_bwd_weight':
    return 'conv2d_bww_split_kernel' if split else 'conv2d_bww_kernel'
  if name in ('sphere_conv_fwd', 'sphere_conv_bn_eval', 'sphere_conv_bwd_data', 'sphere_conv_bwd_weight'):
    # the 3x3 gnomonic layers of the extractor run on the windowed kernels (split-bf16 ones in bf16x6 mode); the integer-table layers
    # that share these labels (32 -> 288 tap products of cost_conv, the stride-2 3x3 layer) on the general gather-and-MAC kernels
    m = re.search(r'\[(\d+)->(\d+) ', label)
    windowed = bool(m) and int(m.group(2)) % 128 == 0 and int(m.group(1)) % 16 == 0
    if name in ('sphere_conv_fwd', 'sphere_conv_bn_eval'):
      return ('sphere_fwd_split_kernel' if split else 'sphere_fwd_win_kernel') if windowed else 'sphere_fwd_kernel'
    if name == 'sphere_conv_bwd_data':
      return 'sphere_bwd_data_split_kernel' if split else ('sphere_bwd_data_adj9_kernel' if windowed else 'sphere_bwd_data_adj_kernel')
    return ('sphere_bww_split_kernel' if split else 'sphere_bww_win_kernel') if windowed else 'sphere_bwd_weight_kernel'
  if name.startswith('grad_sum'):
    return 'sum_n_kernel'  # (N-ary sum of the gradients that meet at a fan-out, functional.FanOutFunction)
  if name in ('head_fwd', 'head_bwd'):
    # one thread per pixel with the logit column in registers when maxdisp/4 is one of the instantiated depths (csrc/head.hip)
    m = re.search(r'\[(\d+)x(\d+)x(\d+)\]', label)  # [D4 x H x W] of the full-resolution output
    fast = (not m) or int(m.group(1)) in (4, 8, 12, 16, 48, 64)
    if name == 'head_fwd':
      return 'head_fwd_fast_kernel' if fast else 'head_fwd_kernel'
    if fast and ((not m) or int(m.group(3)) <= 512):  # one block per image row: per-pixel pass and row sums in one kernel
      return 'head_bwd_pixrows_kernel+head_bwd_cols_kernel'
    return ('head_bwd_pix_fast_kernel' if fast else 'head_bwd_pix_kernel') + '+head_bwd_rows_kernel+head_bwd_cols_kernel'
  if name == 'classif_fwd':
    return 'bn_stats_kernel+classif_fwd_kernel'
  if name == 'classif_bwd':  # (rows that are multiples of 16 bytes: the 16-byte-access kernels of csrc/classif_head.hip)
    return 'classif_bww2_kernel+classif_bwd_reduce_kernel+classif_bwd_apply2_kernel'
  if name == 'smooth_l1_masked':
    return 'smooth_l1_partial_kernel+smooth_l1_final_kernel'
  return {'bn_train_fwd': 'bn_stats_kernel+bn_apply_kernel', 'bn_train_bwd': 'bn_bwd_stats_kernel+bn_bwd_apply_kernel',
          'bn_eval_fwd': 'bn_eval_kernel',
          'cost_volume_fwd': 'cost_volume_fwd_v4', 'cost_volume_bwd': 'cost_volume_bwd_v4',
          'cost_conv_assemble_fwd': 'cost_conv_assemble_fwd_kernel', 'cost_conv_assemble_bwd': 'cost_conv_assemble_bwd_kernel',
          'maxpool2x2_fwd': 'maxpool2_fwd_kernel', 'maxpool2x2_bwd': 'maxpool2_bwd_kernel', 'depth_to_space2': 'shuffle2_kernel',
          'space_to_depth2': 'unshuffle2_kernel', 'conv1x1_sigmoid_fwd': 'head1_fwd_kernel',
          'conv1x1_sigmoid_bwd': 'head1_bwd_kernel+head1_reduce_kernel', 'deconv2x2_gemm': 'conv1x1_kernel', 'deconv2x2_bwd_data': 'conv1x1_kernel',
          'deconv2x2_bwd_weight': 'conv1x1_bww_kernel',
          'conv_stem_fwd': 'stem_fwd_kernel', 'conv_stem_bwd_weight': 'stem_bww_kernel', 'conv1x1_fwd': 'conv1x1_kernel',
          'conv1x1_bwd_data': 'conv1x1_kernel', 'conv1x1_bwd_weight': 'conv1x1_bww_kernel',
          'abs_max': 'abs_max_kernel', 'abs_max_batch': 'abs_max_batch_kernel'}.get(name, name)


def calibrated_traffic(label, batch, conv_arith, on_split=None):
  """HBM bytes per launch of a label from the PMC passes (profiles/traffic.json, see profiles/README.md), or None."""
  try:
    with open(os.path.join(ROOT, 'profiles', 'traffic.json')) as f:
      table = json.load(f)
  except (OSError, ValueError):
    return None
  # (the extractor's kernels see both views: 2 x batch images; the split kernels have their own entries)
  tail = ' bf16x6' if (conv_arith == 'bf16x6' and (on_split if on_split is not None else _on_split_path)(label)) else ''
  for key in ('%s B=%d%s' % (label, batch, tail), '%s B=%d%s' % (label, 2 * batch, tail), label + tail):
    if key in table:
      return table[key].get('hbm_bytes_per_launch')
  return None


def dtype_string(conv_arith):
  """The bench line's `dtype`: the NARROWEST product arithmetic of the timed step first (the driver keeps ~100 characters), then
  where the wider ones are used.  Storage, accumulation, BatchNorm, head and optimizer are fp32 under every setting."""
  if conv_arith == 'f32':
    return 'fp32 storage/accumulate; products: fp32 MFMA everywhere (--conv-arith f32)'
  parts = []
  if CONV3D_S1_F16:
    who = ['stride-1 3x3x3']
    if SPHERE_FWD_F16:
      who.append('spherical')
    if CONV2D_F16:
      who.append('3x3')
    parts.append('fp32 storage/accumulate; products: 2xfp16 split (22-bit, per-tensor 2^k scale, 3 MFMAs) in the %s layers incl. '
                 'gradients (DESIGN 3u/3v)' % ' / '.join(who))
    parts.append('3xbf16 split (24-bit, 6 MFMAs) in the stride-2 / transposed 3x3x3 layers and the polar spherical tiles (3j-3l)')
  else:
    parts.append('fp32 storage/accumulate; products: 3xbf16 split (24-bit, 6 bf16 MFMAs) in the 3x3x3, 3x3 and spherical layers incl. '
                 'gradients (DESIGN 3j-3l)')
  parts.append('fp32 MFMA in the 32->1 heads, 1x1 and 7x7 layers')
  return '; '.join(parts)



def roofline_block(kern, conv_arith, batch, profile_steps, timed_over, on_split=None):
  """The bench line's `roofline` object.  The dominant kernel is the DEVICE KERNEL with the largest total time over all the layer
  shapes it serves -- the first row of `rocprofv3 --stats` for the same command -- priced against the pipe its launches run on:
  achieved = algorithmic flops (bytes) of all its launches / their total duration, i.e. per average launch.  `traffic` is the PMC
  figure of its heaviest layer shape (`traffic_label`).  `by_label` is the (operator, layer shape) label with the largest total
  time, the finer-grained view the per-label table `kernels` is keyed by."""
  steps = max(profile_steps, 1)
  groups = by_kernel(kern, conv_arith, on_split)
  # the dominant DEVICE KERNEL: groups named a+b are operators made of several kernels (BatchNorm backward = statistics + apply), which
  # rocprofv3 --stats lists as the separate rows they are; the largest of those operators is reported beside it (`largest_operator`)
  single = [k for k in groups if '+' not in k] or list(groups)
  kdom = max(single, key=lambda k: groups[k]['total_ms'])
  kop = max(groups, key=lambda k: groups[k]['total_ms'])
  g = groups[kdom]
  g_sec = g['total_ms'] * 1e-3
  heavy = max(g['labels'], key=lambda k: kern[k]['total_ms'])
  _, kpeak, kunit = label_peak(heavy, conv_arith, on_split)
  work = g['bytes'] if g['bound'] == 'hbm' else g['flops']
  dom = max(kern, key=lambda k: kern[k]['total_ms'])
  a = kern[dom]
  bound, peak, unit = label_peak(dom, conv_arith, on_split)
  achieved, per_launch = (a['GBps'], a['bytes_per_call']) if bound == 'hbm' else (a['TFLOPs'], a['flops_per_call'])
  return {'kernel': kdom, 'bound': g['bound'], 'achieved': work / g_sec / (1e9 if g['bound'] == 'hbm' else 1e12), 'peak': kpeak, 'unit': kunit,
          'frac': g['need_s'] / g_sec, 'traffic': calibrated_traffic(heavy, batch, conv_arith, on_split), 'traffic_label': heavy,
          'algorithmic_per_launch': work / max(g['calls'], 1), 'avg_ms': g['total_ms'] / max(g['calls'], 1), 'calls': g['calls'],
          'ms_per_step': g['total_ms'] / steps, 'launches_per_step': g['calls'] / steps, 'labels': g['labels'],
          'selected_by': 'largest total time of one device kernel over all the layer shapes it serves (the first row of rocprofv3 --stats); '
                         'by_label = largest total time of one (operator, layer shape) label',
          'by_label': {'kernel': dom, 'bound': bound, 'achieved': achieved, 'peak': peak, 'unit': unit, 'frac': achieved / peak,
                       'traffic': calibrated_traffic(dom, batch, conv_arith, on_split), 'algorithmic_per_launch': per_launch,
                       'avg_ms': a['avg_ms'], 'calls': a['calls']},
          'largest_operator': {'kernels': kop, 'bound': groups[kop]['bound'], 'frac': groups[kop]['need_s'] / (groups[kop]['total_ms'] * 1e-3),
                               'ms_per_step': groups[kop]['total_ms'] / steps, 'launches_per_step': groups[kop]['calls'] / steps},
          'timed_over': timed_over}


def by_kernel(kern, conv_arith, on_split=None):
  """{device kernel: dict(total_ms, calls = kernel LAUNCHES, flops, bytes, need_s)} over all labels (need_s = time its pipe's peak would need)."""
  out = {}
  for k, v in kern.items():
    bound, peak, unit = label_peak(k, conv_arith, on_split)
    a = out.setdefault(kernel_of(k, conv_arith, on_split), dict(total_ms=0.0, calls=0, flops=0, bytes=0, need_s=0.0, bound=bound, labels=[]))
    a['total_ms'] += v['total_ms']
    a['calls'] += v['calls']  # (every labelled operator of the step is ONE launch of its device kernel -- since round 3 also the 64-channel stride-1 3-D layers)
    a['flops'] += v['flops']
    a['bytes'] += v['bytes']
    a['need_s'] += (v['bytes'] / (peak * 1e9)) if bound == 'hbm' else (v['flops'] / (peak * 1e12))
    a['labels'].append(k)
  return out


def synthetic_batch(B, H, W, maxdisp, device, seed):
  """left = ImageNet-normalised uniform noise; right = left shifted along W + noise; ground truth disparity in
  [0, maxdisp/2] with 5 % NaN (mask of train_disparity.py:195)."""
  g = torch.Generator(device='cpu').manual_seed(seed)
  mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
  std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
  left = (torch.rand(B, 3, H, W, generator=g) - mean) / std
  right = torch.roll(left, -7, 3) + 0.01 * torch.randn(B, 3, H, W, generator=g)
  gt = torch.rand(B, 1, H, W, generator=g) * (maxdisp / 2.0)
  gt[torch.rand(B, 1, H, W, generator=g) < 0.05] = float('nan')
  return left.to(device), right.to(device), gt.to(device)


def usable_cores():
  """Cores this process may actually use: affinity mask, capped by the cgroup CPU quota (a container on a big host)."""
  n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
  try:
    with open('/sys/fs/cgroup/cpu.max') as f:
      quota, period = f.read().split()
    if quota != 'max':
      n = min(n, max(1, int(float(quota) / float(period))))
  except (OSError, ValueError):
    try:
      with open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us') as f, open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as g:
        q, p = int(f.read()), int(g.read())
      if q > 0:
        n = min(n, max(1, q // p))
    except (OSError, ValueError):
      pass
  return n


def cpu_baseline(args):
  """Time the CPU oracle (oracle/mode_ref.py, same arithmetic as the reference on torch's CPU backend) on this host.
  Bounded: one fwd+bwd at BASELINE configs[0] size (Cassini 512x256, D=64, B=1); if that predicts < 45 s for the full
  1024x512 / D=192 pair, the full-size pair is run and reported instead."""
  import recipe
  from oracle import mode_ref
  cores = usable_cores()
  torch.set_num_threads(cores)

  def run(maxdisp, H, W):
    P = recipe.recipe_state(recipe.load_manifest(), 1)
    for k, v in P.items():
      if v.is_floating_point() and 'running' not in k:
        v.requires_grad_(True)
    left, right = recipe.recipe_images(1, H, W, 2)
    gt = recipe.recipe_disparity(1, H, W, 3, maxdisp)
    pos = mode_ref.sphere_position(H // 4, W // 4, 'Cassini')
    t0 = time.time()
    preds = mode_ref.mode_disparity(P, left, right, maxdisp, pos, True)
    mode_ref.training_loss(preds, gt, ~torch.isnan(gt)).backward()
    return time.time() - t0

  run(16, 64, 32)  # warm-up (thread pools, allocator)
  t_small = run(64, 512, 256)
  flop_ratio = 1574.6 / 224.7  # fwd GFLOP per pair, full size vs configs[0]
  full = (args.maxdisp, args.height, args.width) == (192, 1024, 512)
  if full and t_small * flop_ratio < 45.0:
    t_full = run(192, 1024, 512)
    return dict(value=1.0 / t_full, unit='pairs/s', cores=cores, kind='port',
                sample='1 pair fwd+bwd at full size (Cassini 1024x512, D=192), oracle/mode_ref.py on torch CPU, %.1f s' % t_full)
  return dict(value=1.0 / (t_small * flop_ratio), unit='pairs/s', cores=cores, kind='port',
              sample='1 pair fwd+bwd at Cassini 512x256, D=64 (%.1f s), scaled to 1024x512/D=192 by the FLOP ratio %.2f; '
              'oracle/mode_ref.py on torch CPU' % (t_small, flop_ratio))


def cpu_baseline_eval(args):
  """The inference leg's baseline (BASELINE configs[1]): ONE eval forward of the CPU oracle at the bench size, no autograd."""
  import recipe
  from oracle import mode_ref
  cores = usable_cores()
  torch.set_num_threads(cores)
  P = recipe.recipe_state(recipe.load_manifest(), 1)
  pos = mode_ref.sphere_position(args.height // 4, args.width // 4, 'Cassini')
  with torch.no_grad():
    l0, r0 = recipe.recipe_images(1, 64, 32, 2)
    mode_ref.mode_disparity(P, l0, r0, 16, mode_ref.sphere_position(16, 8, 'Cassini'), False)  # warm-up
    left, right = recipe.recipe_images(1, args.height, args.width, 2)
    t0 = time.time()
    mode_ref.mode_disparity(P, left, right, args.maxdisp, pos, False)
    t = time.time() - t0
  return dict(value=1.0 / t, unit='pairs/s', cores=cores, kind='port',
              sample='1 pair eval forward at Cassini %dx%d, D=%d, oracle/mode_ref.py on torch CPU, %.1f s' % (args.height, args.width, args.maxdisp, t))


def cpu_baseline_fusion(args):
  """The fusion leg's baseline: ONE inference forward of oracle/fusion_ref.py (plain torch CPU ops) at the bench size."""
  from oracle import fusion_ref
  import models
  cores = usable_cores()
  torch.set_num_threads(cores)
  torch.manual_seed(0)
  net = models.ModeFusion(1000, [32, 64, 128, 256], {'depth': 12, 'rgb': 12})
  P = {k: v.detach() for k, v in net.state_dict().items()}
  g = torch.Generator().manual_seed(1)
  H, W = args.height, args.width
  mk = lambda n, c, s: [torch.rand(1, c, H, W, generator=g) * s for _ in range(n)]  # noqa: E731
  with torch.no_grad():
    fusion_ref.mode_fusion(P, [t[:, :, :64, :32] for t in mk(6, 1, 50.0)], [t[:, :, :64, :32] for t in mk(6, 1, 1.0)],
                           [t[:, :, :64, :32] for t in mk(4, 3, 1.0)], 1000.0, False)  # warm-up
    d, c, r = mk(6, 1, 50.0), mk(6, 1, 1.0), mk(4, 3, 1.0)
    t0 = time.time()
    fusion_ref.mode_fusion(P, d, c, r, 1000.0, False)
    t = time.time() - t0
  return dict(value=1.0 / t, unit='frames/s', cores=cores, kind='port',
              sample='1 inference forward of ModeFusion(1000,[32,64,128,256]) at %dx%d, oracle/fusion_ref.py on torch CPU, %.1f s' % (H, W, t))


def cpu_baseline_subprocess(args, which='train'):
  """Run the CPU timing in a child process (never touches the GPU) so that a slow host cannot stall the bench line."""
  import subprocess
  cmd = [sys.executable, os.path.abspath(__file__), '--cpu-baseline-only', '--height', str(args.height), '--width', str(args.width),
         '--maxdisp', str(args.maxdisp), '--mode', which]
  try:
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=args.cpu_baseline_timeout)
    for line in reversed(r.stdout.strip().splitlines()):
      if line.startswith('{'):
        return json.loads(line)
    return dict(value=None, unit='pairs/s', cores=usable_cores(), kind='port', sample='failed: ' + r.stderr[-300:])
  except subprocess.TimeoutExpired:
    return dict(value=None, unit='pairs/s', cores=usable_cores(), kind='port',
                sample='timed out after %d s (1 pair fwd+bwd at Cassini 512x256, D=64)' % args.cpu_baseline_timeout)


NOMINAL_SCLK_MHZ = 2400.0  # the engine clock the guide's matrix peaks are quoted at


def sample_clocks(step, fence, device_index, seconds=4.0):
  """Engine clock and socket power WHILE the step replays (the step runs at the socket's power limit, ~10 % under the clock
  the matrix peaks are quoted at).  After the timed region: `seconds` more of back-to-back steps, `rocm-smi --showclocks --showpower`
  polled from a thread meanwhile (a child process per poll; nothing of it touches the timed steps).  None when rocm-smi is not there
  or prints nothing for this device."""
  import re
  import shutil
  import subprocess
  import threading
  smi = shutil.which('rocm-smi') or '/opt/rocm/bin/rocm-smi'
  if not os.path.exists(smi):
    return None
  pat_c = re.compile(r'GPU\[%d\]\s*:\s*sclk clock level:[^(]*\((\d+)Mhz\)' % device_index)
  pat_p = re.compile(r'GPU\[%d\]\s*:\s*Current Socket Graphics Package Power \(W\):\s*([\d.]+)' % device_index)

  def poll_once():
    try:
      r = subprocess.run([smi, '--showclocks', '--showpower'], capture_output=True, text=True, timeout=10)
    except (OSError, subprocess.SubprocessError):
      return None
    c, w = pat_c.search(r.stdout), pat_p.search(r.stdout)
    return (int(c.group(1)), float(w.group(1)) if w else None) if c else None

  samples, stop = [], threading.Event()

  def poll():
    while not stop.is_set():
      v = poll_once()
      if v is None:
        return
      samples.append(v)

  th = threading.Thread(target=poll, daemon=True)
  t0 = time.time()
  for _ in range(5):  # (the governor's time constant: let the load settle before the first sample)
    step()
  fence()
  th.start()
  n = 0
  while time.time() - t0 < seconds or (not samples and th.is_alive() and time.time() - t0 < 4 * seconds):
    for _ in range(5):
      step()
    fence()
    n += 5
  stop.set()
  th.join(timeout=12)
  if len(samples) > 1:
    samples = samples[:-1]  # (the last poll may have straddled the end of the load)
  if not samples:
    return None
  time.sleep(1.0)
  idle = poll_once()
  clk = sorted(c for c, _ in samples)
  pw = sorted(w for _, w in samples if w is not None)
  return {'sclk_mhz_under_step': clk[len(clk) // 2], 'sclk_mhz_range': [clk[0], clk[-1]], 'socket_power_w_under_step': pw[len(pw) // 2] if pw else None,
          'samples': len(samples), 'extra_steps': n + 5, 'sclk_mhz_idle_after': idle[0] if idle else None, 'socket_power_w_idle_after': idle[1] if idle else None,
          'nominal_sclk_mhz': NOMINAL_SCLK_MHZ,
          'method': 'rocm-smi --showclocks --showpower polled from a thread while the step replays back to back AFTER the timed region'}


def eval_b1_leg(net, left, right, args, steps=20, warmup=3):
  """BASELINE configs[1]: eval forward of ONE pair (BatchNorm folded into the convolution kernels, hipGraph replay), timed right
  after the training steps on the same weights: ms per pair, pairs/s, and the dominant kernel label of two profiled eager passes."""
  from mode_hip import profiling
  from mode_hip.graph_step import GraphedStep
  was_training = net.training
  net.eval()

  def fwd():
    with torch.no_grad():
      return net(left, right)

  try:
    for _ in range(warmup):
      fwd()
    torch.cuda.synchronize()
    launch = 'hipGraph replay'
    try:
      graphed = GraphedStep(fwd, (left, right), warmup=1)
      run = graphed.replay
    except Exception as e:
      sys.stderr.write('bench.py: eval_b1 capture failed (%s); eager\n' % e)
      launch, run = 'eager', fwd
    run()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(steps):
      run()
    torch.cuda.synchronize()
    ms = 1e3 * (time.time() - t0) / steps
    out = {'workload': 'eval forward, batch 1 (BASELINE configs[1])', 'ms_per_pair': ms, 'pairs_per_s': 1e3 / ms, 'steps': steps, 'launch': launch}
    # the same forward at the batches that fill the chip: ms per PAIR at 2 and 4 pairs per call says how much of the
    # one-pair time is under-filled launches (one pair = 2 images: half the workgroups of a training launch)
    fill = {'1': round(ms, 3)}
    for nb in (2, 4):
      try:
        lb, rb = torch.cat([left] * nb)[:nb].contiguous(), torch.cat([right] * nb)[:nb].contiguous()

        def fwd_nb():
          with torch.no_grad():
            return net(lb, rb)

        for _ in range(2):
          fwd_nb()
        torch.cuda.synchronize()
        try:
          run_nb = GraphedStep(fwd_nb, (lb, rb), warmup=1).replay
        except Exception:
          run_nb = fwd_nb
        run_nb()
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(8):
          run_nb()
        torch.cuda.synchronize()
        fill[str(nb)] = round(1e3 * (time.time() - t0) / 8 / nb, 3)
        del lb, rb
      except Exception as e:  # (memory: a 4-pair eval forward needs ~9 GB)
        fill[str(nb)] = 'failed: %s' % str(e)[:80]
    out['ms_per_pair_at_batch'] = fill
    if not args.no_kernel_timing:
      profiling.enable(True)
      for _ in range(2):
        fwd()
      torch.cuda.synchronize()
      kern = profiling.summary()
      profiling.enable(False)
      if kern:
        dom = max(kern, key=lambda k: kern[k]['total_ms'])
        bound, peak, unit = label_peak(dom, args.conv_arith)
        ach = kern[dom]['GBps'] if bound == 'hbm' else kern[dom]['TFLOPs']
        out['dominant'] = {'kernel': dom, 'ms_per_pair': kern[dom]['total_ms'] / 2, 'calls_per_pair': kern[dom]['calls'] // 2, 'achieved': ach,
                           'unit': unit, 'frac': ach / peak}
    return out
  finally:
    net.train(was_training)


def rccl_self_test(reducer, backend):
  """World size 1 through the real collective path: init_process_group(backend) on 127.0.0.1, five all-reduces of the flat gradient
  buffer, timed with events on the current stream.  One GPU cannot measure scaling, but it can prove that RCCL initialises and that
  the buffer the 8-rank run will reduce is acceptable to it -- so that the first multi-GPU run does not discover either."""
  import socket
  out = {'backend': 'rccl' if backend == 'nccl' else backend, 'ranks': 1, 'self_test': True,
         'op': 'all_reduce(sum) of the flat gradient buffer', 'bytes': reducer.flat.numel() * 4}
  try:
    with socket.socket() as sk:
      sk.bind(('127.0.0.1', 0))
      port = sk.getsockname()[1]
    dist.init_process_group(backend, init_method='tcp://127.0.0.1:%d' % port, rank=0, world_size=1)
    try:
      before = reducer.flat.clone()
      dist.all_reduce(reducer.flat, op=dist.ReduceOp.SUM)  # (also creates the communicator)
      torch.cuda.synchronize()
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      for _ in range(5):
        dist.all_reduce(reducer.flat, op=dist.ReduceOp.SUM)
      e1.record()
      torch.cuda.synchronize()
      out['avg_ms_rank0'] = round(e0.elapsed_time(e1) / 5, 4)
      out['ok'] = bool(torch.equal(before, reducer.flat))  # a sum over one rank is the identity
    finally:
      dist.destroy_process_group()
  except Exception as e:  # reported, never fatal: the measurement above is already complete
    out['ok'] = False
    out['error'] = '%s: %s' % (type(e).__name__, str(e)[:300])
  return out


def fusion_main(args):
  """`--mode fusion`: inference forward of the fusion network as train_fusion.py:64 builds it -- ModeFusion(1000, [32, 64, 128, 256],
  {'depth': 12, 'rgb': 12}) on 6 depth + 6 confidence maps and 4 RGB views of --height x --width (the second half of BASELINE
  configs[4]) -- on ONE GPU, BatchNorm folded into the convolution kernels, hipGraph replay.  Same line format as the headline bench:
  frames/s, the dominant kernel's roofline entry, the CPU oracle's time beside it."""
  assert args.gpus == 1, 'the fusion leg is a one-GPU inference benchmark'
  assert torch.cuda.is_available(), 'bench.py needs a GPU (the product has no CPU path)'
  dev = torch.device('cuda', 0)
  torch.cuda.set_device(0)
  import models
  import mode_hip
  mode_hip.lib()
  from mode_hip import functional as HF, no_vendor, profiling
  from mode_hip.graph_step import GraphedStep
  HF.set_conv_arith(args.conv_arith)
  B, H, W = args.batch, args.height, args.width  # (default: one frame per call, configs[4]'s share of a GPU)
  torch.manual_seed(0)
  net = models.ModeFusion(1000, [32, 64, 128, 256], {'depth': 12, 'rgb': 12}).to(dev)
  g = torch.Generator().manual_seed(1)
  depthes = [(torch.rand(B, 1, H, W, generator=g) * 50).to(dev) for _ in range(6)]
  confs = [torch.rand(B, 1, H, W, generator=g).to(dev) for _ in range(6)]
  rgbs = [torch.rand(B, 3, H, W, generator=g).to(dev) for _ in range(4)]
  net.train()
  with torch.no_grad():  # calibrate the running statistics (fresh ones would let the activations grow layer by layer)
    for _ in range(2):
      net(depthes, confs, rgbs)
  net.eval()

  def fwd():
    with torch.no_grad():
      return net(depthes, confs, rgbs)

  with no_vendor.no_vendor_arithmetic() as guard:
    out0 = fwd()
  assert bool(torch.isfinite(out0).all())
  for _ in range(args.warmup):
    fwd()
  torch.cuda.synchronize()
  launch, run = 'hipGraph replay', None
  try:
    graphed = GraphedStep(fwd, tuple(depthes + confs + rgbs), warmup=1)
    run = graphed.replay
  except Exception as e:
    sys.stderr.write('bench.py: fusion capture failed (%s); eager\n' % e)
    launch, run = 'eager', fwd
  run()
  torch.cuda.synchronize()
  t0 = time.time()
  for _ in range(args.steps):
    run()
  torch.cuda.synchronize()
  elapsed = time.time() - t0
  kern = None
  if not args.no_kernel_timing:
    profiling.enable(True)
    for _ in range(args.profile_steps):
      fwd()
    torch.cuda.synchronize()
    kern = profiling.summary()
    profiling.enable(False)
  out = {
      'metric': 'ModeFusion inference forward frames/sec (%dx%d, 6 depth+confidence views, 4 RGB)' % (W, H),
      'value': B * args.steps / elapsed, 'unit': 'frames/s', 'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup,
      'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
      'dtype': 'fp32 storage/accumulate; products: 3xbf16 split (24-bit, 6 MFMAs) in the 3x3 layers with 16k input channels, fp32 MFMA in the '
               '12-channel input layers and the 2x2 transposed convolutions' if args.conv_arith == 'bf16x6' else 'fp32 storage/accumulate; fp32 MFMA',
      'data': 'synthetic', 'peak_mem_gb': round(torch.cuda.max_memory_allocated() / 2**30, 2),
      'config': {'workload': 'ModeFusion(1000,[32,64,128,256],depth 12,rgb 12) eval forward, %d frame(s)/call, %dx%d (BASELINE configs[4], '
                             'fusion half; train_fusion.py:64)' % (B, H, W),
                 'global_batch': B, 'conv_arith': args.conv_arith, 'launch': launch, 'init': 'torch.manual_seed(0) constructor init, BatchNorm '
                 'running statistics from two training-mode forwards',
                 'vendor_guard': 'first forward ran under mode_hip.no_vendor (%d aten ops seen, none of them vendor arithmetic)' % guard.seen},
  }
  if kern:
    out['roofline'] = roofline_block(kern, args.conv_arith, B, args.profile_steps, '%d eager forwards after the timed region' % args.profile_steps)
    out['kernels'] = {k: {'calls': v['calls'], 'avg_ms': round(v['avg_ms'], 4), 'GBps': round(v['GBps'], 1), 'TFLOPs': round(v['TFLOPs'], 2)}
                      for k, v in kern.items()}
    conv = [v for k, v in kern.items() if k.startswith('conv2d_')]
    if conv:
      out['roofline']['conv3x3_mfma_frac'] = round(blended_mfma_fraction(kern, args.conv_arith, prefixes=('conv2d_',)), 4)
      out['roofline']['conv3x3_ms_per_forward'] = round(sum(v['total_ms'] for v in conv) / max(args.profile_steps, 1), 3)
  else:
    out['roofline'] = None
  if not args.no_cpu_baseline:
    out['cpu_baseline'] = cpu_baseline_subprocess(args, 'fusion')
  print(json.dumps(out))


def main():
  args = parse()
  if args.cpu_baseline_only:
    print(json.dumps({'train': cpu_baseline, 'eval': cpu_baseline_eval, 'fusion': cpu_baseline_fusion}[args.mode](args)))
    return
  if args.mode == 'fusion':
    return fusion_main(args)
  if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
    sys.exit(launch_ranks(args))  # before anything touches the GPU in this process
  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  if world != args.gpus:
    raise SystemExit('bench.py: --gpus %d but the launcher started %d ranks (WORLD_SIZE)' % (args.gpus, world))
  if world > 1:
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    dist.init_process_group(args.dist_backend, rank=rank, world_size=world)
  assert torch.cuda.is_available(), 'bench.py needs a GPU (the product has no CPU path)'
  local_rank %= torch.cuda.device_count()
  torch.cuda.set_device(local_rank)
  dev = torch.device('cuda', local_rank)

  import models
  from mode_hip import data_parallel, profiling
  import mode_hip
  mode_hip.lib()  # fail loudly if the native library is missing
  from mode_hip import functional as HF
  HF.set_conv_arith(args.conv_arith)
  HF.CONV3D_BN_STATS = bool(args.fused_bn_stats)
  HF.CLASSIF_FUSED = not args.no_fused_classif
  HF.GRAD_CARRIERS = not args.no_grad_carriers
  global CONV3D_S1_F16, SPHERE_FWD_F16, CONV2D_F16, CONV3D_EVAL_F16
  CONV3D_EVAL_F16 = HF.CONV3D_EVAL_F16 = not args.no_eval_f16 and args.conv_arith == 'bf16x6'
  HF.CONV2D_EVAL_F16 = CONV3D_EVAL_F16
  HF.CONV2D_F16 = not args.no_conv2d_f16
  CONV2D_F16 = HF.CONV2D_F16 and args.conv_arith == 'bf16x6' and args.mode == 'train'
  CONV3D_S1_F16 = HF.CONV3D_S1_F16 = not args.no_conv3d_f16 and args.conv_arith == 'bf16x6'
  HF.SPHERE_FWD_F16 = HF.SPHERE_BWD_F16 = not args.no_sphere_f16
  SPHERE_FWD_F16 = HF.SPHERE_FWD_F16 and args.conv_arith == 'bf16x6' and args.mode == 'train'

  torch.backends.cudnn.benchmark = bool(args.vendor_autotune)
  torch.manual_seed(0)
  net = models.ModeDisparity(args.maxdisp, 'Sphere', args.height, args.width, 'Cassini').to(dev)
  if args.init == 'recipe':
    # weights from the 8(c) recipe (tests/golden/recipe.py: every tensor of the state_dict drawn from
    # numpy.random.RandomState(seed) in key order with per-kind scales) -- the state the parity fixtures use, reproducible without torch's RNG
    import recipe
    net.load_state_dict(recipe.recipe_state(recipe.load_manifest(), 1))
  reducer = data_parallel.GradAllReducer(net)
  reducer.broadcast_parameters(net)
  try:  # same update rule as train_disparity.py:287 (Adam, lr 1e-3, betas (0.9, 0.999)); single-kernel implementation
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, betas=(0.9, 0.999), fused=True)
  except (TypeError, RuntimeError):
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, betas=(0.9, 0.999))
  left, right, gt = synthetic_batch(args.batch, args.height, args.width, args.maxdisp, dev, seed=1234 + rank)
  # The static inputs of the (graph-replayed) step: images, ground truth WITH its NaNs (the mask is derived inside the step) and
  # the global valid-pixel count -- a property of the batch that needs a collective, so it is computed when a batch is loaded
  # into a buffer the captured step reads (GraphedStep.load(left, right, gt, count)); nothing about the batch is baked into the
  # graph as a constant.
  count = data_parallel.global_valid_count(~torch.isnan(gt))

  def fwd_bwd():
    reducer.zero_grad()
    if not args.no_fused_loss:  # the same loss, formed next to the heads (ModeDisparity.forward_loss; tests/test_gpu_steps.py pins both forms)
      loss, _ = net.forward_loss(left, right, gt, count=count)
    else:
      mask = ~torch.isnan(gt)
      gt0 = torch.nan_to_num(gt)
      o1, o2, o3 = net(left, right)
      loss = 0
      for wgt, o in ((0.5, o1), (0.7, o2), (1.0, o3)):
        loss = loss + wgt * data_parallel.global_masked_mean(F.smooth_l1_loss(o, gt0, reduction='none'), mask, count=count)
    loss.backward()
    return loss

  def eval_fwd():
    with torch.no_grad():
      return net(left, right)

  if args.mode == 'train':
    net.train()
    body = fwd_bwd
  else:
    net.eval()
    body = eval_fwd

  ar_events = []

  def finish():
    if args.mode == 'train':
      if world > 1:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        reducer.all_reduce()
        e1.record()
        ar_events.append((e0, e1))
      opt.step()

  def eager_step():
    body()
    finish()

  def fence():
    torch.cuda.synchronize()
    if world > 1:
      dist.barrier()
    torch.cuda.synchronize()

  # The first eager step runs under a guard that raises on any vendor-library arithmetic (convolution, BatchNorm, GEMM, softmax,
  # interpolation) reached from the step: the host-side helpers keep silent exits to the torch module for shapes the kernels do not
  # take (models/stage3d.conv3 -> conv(x), bn_act_torch), and a timed step must not be one of those.
  from mode_hip import no_vendor
  with no_vendor.no_vendor_arithmetic() as guard:
    eager_step()
  fence()
  for _ in range(args.warmup):
    eager_step()
  fence()
  step = eager_step
  if args.launch == 'graph':
    from mode_hip.graph_step import GraphedStep
    try:
      graphed = GraphedStep(body, (left, right, gt, count), warmup=1)

      def step():
        graphed.replay()
        finish()

      step()  # first replay uploads the graph; not timed
      fence()
    except Exception as e:  # capture is an optimisation of the launch path only: report the eager rate rather than nothing
      sys.stderr.write('bench.py: hipGraph capture failed (%s: %s); timing the eager step instead\n' % (type(e).__name__, e))
      torch.cuda.synchronize()
      args.launch = 'eager'
      step = eager_step
      step()
      fence()

  # Timed region: exactly K steps.  With --launch graph the per-kernel events cannot sit inside the replayed graph, so the
  # per-kernel (roofline) timing is taken over --profile-steps eager steps of the same workload right after the timed
  # region; with --launch eager the events are recorded inside the timed region itself.
  profiling.enable(args.launch == 'eager' and not args.no_kernel_timing)
  del ar_events[:]
  t0 = time.time()
  for _ in range(args.steps):
    step()
  fence()
  elapsed = time.time() - t0
  allreduce_ms = sum(a.elapsed_time(b) for a, b in ar_events) / max(len(ar_events), 1) if ar_events else None
  clocks = None
  if world == 1 and not args.no_clock_sample:
    try:
      clocks = sample_clocks(step, fence, dev.index if dev.index is not None else 0)
    except Exception as e:  # (a reported extra, never a reason to lose the bench line)
      sys.stderr.write('bench.py: clock sampling failed (%s: %s)\n' % (type(e).__name__, e))
  if args.launch == 'graph' and not args.no_kernel_timing:
    eager_step()  # re-warm the eager allocator pool (the replayed steps lived in the graph's private pool): an allocation
    fence()       # that falls through to hipMalloc stalls the stream between the two events of a region
    profiling.enable(True)
    for _ in range(args.profile_steps):
      eager_step()
    fence()
    if net.fold_cost_volume and args.profile_steps:
      # the step no longer builds the cost volume (HF.cost_conv); the a9 kernel remains part of the operator API and its
      # north_star target is measured on the step's shapes, standalone
      from mode_hip import functional as HF
      fr = torch.randn(args.batch, 32, args.height // 4, args.width // 4, device=dev)
      ft = torch.randn_like(fr)
      profiling.ENABLED = False  # (not enable(): that would drop the step's records)
      HF.cost_volume_fwd(fr, ft, args.maxdisp // 4)  # allocator warm-up for the 400 MB/sample result, untimed
      fence()
      profiling.ENABLED = True
      for _ in range(3):
        HF.cost_volume_fwd(fr, ft, args.maxdisp // 4)
      fence()
      del fr, ft
  kern = profiling.summary()
  if args.mode == 'train':
    # sanity of the steps just timed: every parameter still finite after the optimizer updates (one reduction, after the timing)
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    if not bool(torch.isfinite(flat).all()):
      raise RuntimeError('bench.py: non-finite parameters after %d training steps' % (args.warmup + args.steps))
  profiling.enable(False)
  eval_b1 = None
  if world == 1 and args.mode == 'train' and not args.no_eval_b1:
    eval_b1 = eval_b1_leg(net, left[:1], right[:1], args)
  self_test = None
  if world == 1 and args.mode == 'train' and not args.no_collective_self_test:
    self_test = rccl_self_test(reducer, args.dist_backend)
  rank_ms = [1e3 * elapsed / args.steps]
  if world > 1:
    t = torch.zeros(world, device=dev, dtype=torch.float64)
    t[rank] = elapsed
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    rank_ms = [1e3 * float(v) / args.steps for v in t]
    elapsed = float(t.max())

  if rank == 0:
    pairs = args.batch * world * args.steps
    out = {
        'metric': 'ERP stereo pairs/sec (512x1024, 192 disp) fwd+bwd' if args.mode == 'train' else 'ERP stereo pairs/sec (512x1024, 192 disp) fwd',
        'value': pairs / elapsed,
        'unit': 'pairs/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': 1e3 * elapsed / args.steps,
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': dtype_string(args.conv_arith),
        'data': 'synthetic',
        'peak_mem_gb': round(torch.cuda.max_memory_allocated() / 2**30, 2),
        'per_gpu_value': pairs / elapsed / world,
        'rank_ms_per_step': [round(v, 3) for v in rank_ms],
        'collective': self_test if world == 1 else {'backend': 'rccl' if args.dist_backend == 'nccl' else args.dist_backend, 'ranks': world,
                                               'op': 'all_reduce(sum) of the flat gradient buffer, once per step',
                                               'bytes': reducer.flat.numel() * 4,
                                               'avg_ms_rank0': None if allreduce_ms is None else round(allreduce_ms, 3)},
        'config': {
            'workload': 'ModeDisparity(%d,Sphere,%dx%d Cassini) %s step, batch %d/GPU (BASELINE configs[%d])' %
                        (args.maxdisp, args.height, args.width, 'fwd+bwd+Adam' if args.mode == 'train' else 'eval fwd',
                         args.batch, 2 if world == 1 else 3),
            'global_batch': args.batch * world,
            'conv_arith': args.conv_arith,
            'init': 'recipe state, seed 1 (tests/golden/recipe.py)' if args.init == 'recipe' else 'torch.manual_seed(0) constructor init',
            'parallelism': 'dp%d' % world,
            'cost_volume': ('folded into dres0[0][0] (cost_conv: 18 partial 2-D products + assembly kernel); the volume is not built, '
                            'targets.cost_volume_fwd_hbm_frac times the a9 kernel standalone') if net.fold_cost_volume
                           else 'built (mode_cost_volume_fwd)',
            'launch': 'hipGraph replay of zero-grad+forward+loss+backward, then all-reduce and fused Adam' if args.launch == 'graph'
                      else 'eager (one launch per kernel)',
            'roofline_timing': ('per-kernel HIP events over %d eager steps run right AFTER the timed region (events cannot sit inside a '
                                'replayed hipGraph); `value` / `ms_per_step` come from the %d timed replays only' %
                                (args.profile_steps, args.steps)) if args.launch == 'graph'
                               else 'per-kernel HIP events inside the timed region',
        },
    }
    if clocks:
      out['clocks'] = clocks
    if kern:
      out['roofline'] = roofline_block(kern, args.conv_arith, args.batch, args.profile_steps,
                                       ('%d eager steps after the timed region (the timed steps replay a hipGraph)' % args.profile_steps)
                                       if args.launch == 'graph' else 'the timed region')
      # north_star targets: HBM fraction of the cost-volume build, MFMA fraction of the whole 3D regulariser -- every label against
      # the pipe it runs on (fp32 MFMA 157.3, or bf16 / 6 = 416.7 for the split kernels), time-weighted: never above 1
      cv = kern.get('cost_volume_fwd')
      k3 = [v for k, v in kern.items() if k.startswith(('conv3d_', 'deconv3d_'))]
      cca = [v for k, v in kern.items() if k.startswith('cost_conv_assemble_fwd')]
      out['targets'] = {
          # the a9 operator (mode_cost_volume_fwd), timed STANDALONE after the step at the step's shapes: the product's model no longer
          # builds the volume ...
          'cost_volume_fwd_hbm_frac': round(cv['GBps'] / HBM_PEAK_GBPS, 4) if cv else None,
          'cost_volume_fwd_measured_on': 'mode_cost_volume_fwd launched standalone after the timed region (not part of the step)',
          # ... it runs the folded form, whose HBM-bound assembly pass IS in the step:
          'cost_conv_assemble_fwd_hbm_frac_in_step': round(sum(v['bytes'] for v in cca) / (sum(v['total_ms'] for v in cca) * 1e6) / HBM_PEAK_GBPS, 4) if cca else None,
          # matrix-bound layers of the regulariser (Conv3d / ConvTranspose3d with >= 32 output channels), each against ITS pipe; the
          # single-channel classifier convolutions are HBM-bound and priced in `classifier_heads_hbm_frac`
          'regulariser3d_mfma_frac': round(blended_mfma_fraction(kern, args.conv_arith), 4) if k3 else None,
          'regulariser3d_tflops_fp32_equivalent': round(sum(v['flops'] for v in k3) / (sum(v['total_ms'] for v in k3) * 1e9), 2) if k3 else None,
      }
      ch = [v for k, v in kern.items() if k.startswith(('classif_fwd', 'classif_bwd'))]
      if ch:
        out['targets']['classifier_heads_hbm_frac'] = round(sum(v['bytes'] for v in ch) / (sum(v['total_ms'] for v in ch) * 1e6) / HBM_PEAK_GBPS, 4)
      out['roofline']['targets'] = out['targets']
      for k, v in out['targets'].items():  # (and as scalar keys of `roofline`: the driver's parser keeps scalars, not nested objects)
        if isinstance(v, (int, float)) or v is None:
          out['roofline'][k] = v
      if clocks and out['roofline'].get('bound') == 'mfma':
        # the same fraction against the matrix peak at the clock the chip SUSTAINS under this step (peak scales with the engine clock)
        out['roofline']['sclk_mhz_under_step'] = clocks['sclk_mhz_under_step']
        out['roofline']['frac_at_sustained_clock'] = round(out['roofline']['frac'] * NOMINAL_SCLK_MHZ / clocks['sclk_mhz_under_step'], 4)
      out['config']['vendor_guard'] = 'first eager step ran under mode_hip.no_vendor (%d aten ops seen, none of them vendor arithmetic)' % guard.seen
      out['kernels'] = {k: {'calls': v['calls'], 'avg_ms': round(v['avg_ms'], 4), 'GBps': round(v['GBps'], 1),
                            'TFLOPs': round(v['TFLOPs'], 2)} for k, v in kern.items()}
    else:
      out['roofline'] = None
    if eval_b1 is not None:
      if not args.no_cpu_baseline:  # (the inference leg's own baseline, one eval forward of the CPU oracle)
        eval_b1['cpu_baseline'] = cpu_baseline_subprocess(args, 'eval')
      out['eval_b1'] = eval_b1
    if args.value_1gpu:
      out['scaling_vs_1gpu'] = {'value_1gpu': args.value_1gpu, 'efficiency': out['value'] / (world * args.value_1gpu)}
    if world == 1 and not args.no_cpu_baseline:
      out['cpu_baseline'] = cpu_baseline_subprocess(args, 'eval' if args.mode == 'eval' else 'train')  # (the metric's own pass)
    print(json.dumps(out))
  if world > 1:
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
