"""Hit rate of the eager-vs-graph bit-equality of the two-rank step (tests/test_gpu_two_ranks.py): runs the pair N times, each with
fresh processes exactly as the test does, and prints a per-tensor report for every mismatch.   python tools/two_rank_pair_loop.py [N]"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'mode-2022_amd'), os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')):
  if p not in sys.path:
    sys.path.insert(0, p)

import torch  # noqa: E402


def main():
  import test_gpu_two_ranks as T
  n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
  ref = None
  bad = 0
  with tempfile.TemporaryDirectory() as tmp:
    for i in range(n):
      for launch in ('eager', 'graph'):
        r = T._run_ranks(tmp, 32, 128, 64, launch)
        cur = {'flat': r[0]['flat'].clone(), 'l0': r[0]['local'].clone(), 'l1': r[1]['local'].clone()}
        if ref is None:
          ref, names, shapes = cur, r[0]['names'], r[0]['shapes']
          continue
        msgs = [k + ': ' + T._per_tensor_report(names, shapes, ref[k], cur[k]) for k in ('l0', 'l1', 'flat') if not torch.equal(ref[k], cur[k])]
        if msgs:
          bad += 1
          print('run %d (%s) differs from run 0 (eager):\n' % (i, launch) + '\n'.join(msgs), flush=True)
        else:
          print('run %d (%s): bit-identical to run 0' % (i, launch), flush=True)
  print('%d of %d runs differ from the first' % (bad, 2 * n - 1))


if __name__ == '__main__':
  main()
