"""Instruction mix of the loops of one kernel in a hipcc -S listing.

  python tools/isa_mix.py file.s <kernel-name-substring> [min-mfma-per-loop]

For every backward branch (loop) of the kernel: the number of matrix, plain vector, scalar, LDS, vector-memory, wait and barrier
instructions in the loop body, and the vector instructions by mnemonic.  What the PMC counters give per launch (SQ_INSTS_VALU per
SQ_INSTS_MFMA), this gives per loop -- where the vector work of a kernel sits and what it is made of."""
import collections
import re
import sys


def classify(op):
  if op.startswith('v_mfma') or op.startswith('v_smfma'):
    return 'mfma'
  if op.startswith('v_'):
    return 'valu'
  if op.startswith('ds_'):
    return 'lds'
  if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
    return 'vmem'
  if op.startswith('s_waitcnt'):
    return 'wait'
  if op.startswith('s_barrier'):
    return 'barrier'
  if op.startswith('s_'):
    return 'salu'
  return 'other'


def main():
  path, name = sys.argv[1], sys.argv[2]
  min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 1
  lines = open(path).read().split('\n')
  start = None
  for i, ln in enumerate(lines):
    if name in ln.split(':')[0] and ':' in ln and ln[:1] not in ('.', '\t', ' ', ';'):
      start = i
      print('kernel', ln[:160])
      break
  assert start is not None, 'kernel not found'
  end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
  body = lines[start:end]
  labels = {}
  insts = []  # (op, text)
  for ln in body:
    s = ln.strip()
    if not s or s.startswith(';') or (s.startswith('.') and not re.match(r'^[.\w$]+:', s)):
      continue
    if re.match(r'^[.\w$]+:', s):
      labels[s.split(':')[0]] = len(insts)
      continue
    op = s.split()[0]
    insts.append((op, s))
  print('instructions in kernel: %d' % len(insts), dict(collections.Counter(classify(o) for o, _ in insts)))
  for idx, (op, s) in enumerate(insts):
    if op.startswith('s_cbranch') or op == 's_branch':
      tgt = s.split()[-1]
      if tgt in labels and labels[tgt] <= idx:
        seg = insts[labels[tgt]:idx + 1]
        c = collections.Counter(classify(o) for o, _ in seg)
        if c['mfma'] < min_mfma:
          continue
        print('\nloop %s: %d instructions  ' % (tgt, len(seg)), dict(c))
        if c['mfma']:
          print('  per MFMA: valu %.2f  salu %.2f  lds %.2f  vmem %.2f' % tuple(c[k] / c['mfma'] for k in ('valu', 'salu', 'lds', 'vmem')))
        v = collections.Counter(o for o, _ in seg if classify(o) in ('valu',))
        print('  valu:', ', '.join('%s %d' % kv for kv in v.most_common(30)))
        l = collections.Counter(o for o, _ in seg if classify(o) in ('lds', 'vmem'))
        print('  mem: ', ', '.join('%s %d' % kv for kv in l.most_common(12)))


if __name__ == '__main__':
  main()
