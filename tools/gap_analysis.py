#!/usr/bin/env python3
"""GPU idle time between kernels from a rocprofv3 --kernel-trace CSV.
usage: python tools/gap_analysis.py <kernel_trace.csv> [n_last_kernels]"""
import collections
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
  for r in csv.DictReader(f):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
n = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 3
rows = rows[-n:]
span = rows[-1][1] - rows[0][0]
busy = sum(e - s for s, e, _ in rows)
print('kernels %d  span %.2f ms  busy %.2f ms  idle %.2f ms' % (len(rows), span / 1e6, busy / 1e6, (span - busy) / 1e6))
gaps = collections.defaultdict(lambda: [0, 0])
hist = collections.Counter()
last_end = rows[0][1]
for (s, e, name), (ps, pe, pname) in zip(rows[1:], rows[:-1]):
  g = max(0, s - max(last_end, pe))
  last_end = max(last_end, e)
  key = name.split('(')[0][-60:]
  gaps[key][0] += g
  gaps[key][1] += 1
  hist[min(int(g / 1000), 100)] += 1
print('gap histogram (us: count):', sorted(hist.items())[:40])
print('idle attributed to the kernel that FOLLOWS the gap:')
for k, (g, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:25]:
  print('  %-62s n=%5d  total %8.2f ms  avg %7.1f us' % (k, c, g / 1e6, g / c / 1e3))
