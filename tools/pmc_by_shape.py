#!/usr/bin/env python3
"""Per-kernel, per-launch-shape means of the PMC passes of tools/pmc_trip.sh (the shapes of a kernel are told apart by grid size).
usage: python tools/pmc_by_shape.py gpurun_out/<tag> [name filter]"""
import collections
import csv
import glob
import os
import sys

root, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else '')
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, '*', '*counter_collection.csv')):
  with open(f) as fh:
    for r in csv.DictReader(fh):
      name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:48]
      if flt and flt not in name:
        continue
      agg[(name, int(r['Grid_Size']), int(r['Workgroup_Size']))][r['Counter_Name']].append(float(r['Counter_Value']))
for (name, grid, wg), c in sorted(agg.items()):
  print('%-48s grid %9d wg %4d' % (name, grid, wg))
  for k in sorted(c):
    v = c[k]
    print('    %-28s n=%3d mean=%16.1f' % (k, len(v), sum(v) / len(v)))
