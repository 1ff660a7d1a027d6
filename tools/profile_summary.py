#!/usr/bin/env python3
"""Per-step summary of a rocprofv3 kernel-stats csv of bench.py: groups and top kernels.
usage: python tools/profile_summary.py <bench_kernel_stats.csv> [top N]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ring = [int(r['Calls']) for r in rows if 'conv3d_bwd_weight_ring_kernel' in r['Name'] or 'conv3d_bww_split_kernel' in r['Name']]
steps = (ring[0] / 12.0) if ring else 1.0  # 12 stride-1 3-D weight gradients with Co > 1 per step (13 without the cost-conv fusion: pass the step count as argv[3] then)
if len(sys.argv) > 3:
  steps = float(sys.argv[3])
skip = ('naive_conv', 'kernel_batched_gemm_xdlops_bwd_weight', 'kernel_grouped_conv')


def grp(n):
  if 'sphere' in n or 'transpose_planes' in n or 'reduce_gw_win' in n:
    return 'sphere'
  if 'cost_conv' in n:
    return 'cost_conv assembly'
  if 'conv2d_' in n or 'reduce_gw2d' in n or 'pack_w2d' in n or 'conv1x1' in n or 'stem_' in n or 'pack_w1' in n or 'pack_w_stem' in n or \
      'zero_insert2' in n or 'reduce_slices' in n:
    return 'conv2d (own)'
  if 'conv3d' in n or 'deconv3d' in n or 'reduce_gw3d' in n or 'pack_w3d' in n:
    return 'conv3d'
  if 'bn_' in n[:60]:
    return 'bn'
  if 'head_' in n:
    return 'head'
  if 'cost_volume' in n:
    return 'cost volume'
  if 'Cijk' in n:
    return 'vendor gemm'
  if 'miopen' in n.lower() or 'igemm' in n or 'batched_transpose' in n or 'SubTensor' in n:
    return 'vendor conv2d'
  if 'elementwise' in n or 'at::native' in n:
    return 'torch elementwise'
  return 'other'


groups, tot = {}, 0.0
for r in rows:
  if any(k in r['Name'] for k in skip):
    continue
  t = float(r['TotalDurationNs']) / 1e6 / steps
  tot += t
  groups[grp(r['Name'])] = groups.get(grp(r['Name']), 0) + t
print('steps in file: %.2f   kernel time per step: %.1f ms' % (steps, tot))
for k, v in sorted(groups.items(), key=lambda kv: -kv[1]):
  print('  %-20s %6.2f ms' % (k, v))
print()
for r in rows[:top]:
  if any(k in r['Name'] for k in skip):
    continue
  print('%-100s %5d %8.3f ms/step' % (r['Name'][:100], int(r['Calls']), float(r['TotalDurationNs']) / 1e6 / steps))
