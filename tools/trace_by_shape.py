#!/usr/bin/env python3
"""Per (kernel, grid size) launch counts and average durations from a rocprofv3 --kernel-trace csv: a kernel that serves several
layer shapes (the ring weight-gradient kernel: three volumes) has ONE row in --stats; this tells the shapes apart.
usage: python tools/trace_by_shape.py <bench_kernel_trace.csv> [name filter] """
import collections
import csv
import sys

agg = collections.defaultdict(list)
flt = sys.argv[2] if len(sys.argv) > 2 else ''
for r in csv.DictReader(open(sys.argv[1])):
  n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
  if flt in n:
    agg[(n.split('(')[0][:64], int(r.get('Grid_Size') or r.get('Grid_Size_X')), int(r.get('Workgroup_Size') or r.get('Workgroup_Size_X')))].append(
        int(r['End_Timestamp']) - int(r['Start_Timestamp']))
rows = sorted(agg.items(), key=lambda kv: -sum(kv[1]))
print('%-64s %10s %5s %6s %12s %12s' % ('kernel', 'grid', 'wg', 'calls', 'avg us', 'total ms'))
for (n, g, w), v in rows:
  print('%-64s %10d %5d %6d %12.1f %12.3f' % (n, g, w, len(v), sum(v) / len(v) / 1e3, sum(v) / 1e6))
