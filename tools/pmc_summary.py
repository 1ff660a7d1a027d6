#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter CSVs per kernel name (mean per dispatch)."""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
  for r in csv.DictReader(open(f)):
    name = r.get('Kernel_Name', '').replace('(anonymous namespace)::', '')[:60]
    agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
for name in sorted(agg):
  if not any(k in name for k in ('sphere', 'conv3d', 'deconv', 'head', 'bn_', 'cost_volume', 'reduce', 'pack')):
    continue
  print(name)
  for c, v in sorted(agg[name].items()):
    print('    %-28s n=%4d mean=%14.1f' % (c, len(v), sum(v) / len(v)))
