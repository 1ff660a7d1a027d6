#!/usr/bin/env python3
"""Run the single-GPU BASELINE.json configs once each and print pairs/s + peak memory (GPU box).

  configs[1]: 512x1024 ERP (Cassini 1024x512), 192 disp, batch 1, eval forward
  configs[2]: same, batch 2, fwd+bwd+Adam                       (= bench.py default)
  configs[4]: 1024x2048 ERP (Cassini 2048x1024), 256 disp, one sample per GPU, fwd+bwd+Adam (the per-GPU share of config 5)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RUNS = [
    ('configs[1] eval fwd  1024x512 D=192 B=1', ['--mode', 'eval', '--batch', '1']),
    ('configs[2] train     1024x512 D=192 B=2', ['--mode', 'train', '--batch', '2']),
    ('configs[4] train    2048x1024 D=256 B=1', ['--mode', 'train', '--batch', '1', '--height', '2048', '--width', '1024', '--maxdisp', '256']),
]


def main():
  import json
  for name, extra in RUNS:
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '3', '--warmup', '2', '--no-cpu-baseline'] + extra
    r = subprocess.run(cmd, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith('{')]
    if not line:
      print('%-46s FAILED\n%s' % (name, r.stderr[-600:]))
      continue
    d = json.loads(line[-1])
    print('%-46s %8.2f pairs/s  %9.2f ms/step  peak %.1f GB  dominant %s (%.0f%% of %s peak)' %
          (name, d['value'], d['ms_per_step'], d.get('peak_mem_gb', 0.0), d['roofline']['kernel'], 100 * d['roofline']['frac'],
           d['roofline']['bound']), flush=True)


if __name__ == '__main__':
  main()
