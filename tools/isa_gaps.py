"""Other instructions per MFMA gap in the innermost MFMA loop of one kernel of a `hipcc -S` listing.

  python tools/isa_gaps.py file.s <mangled-kernel-name-prefix>

One wave per SIMD hides about five single-issue instructions beside a v_mfma_f32_32x32x16 (MI355X_MICROARCH.md); what matters is how
the loop's other instructions are DISTRIBUTED over the gaps between consecutive MFMAs -- a greedy group pattern fills the first gaps
of a stage and leaves the rest empty.  Prints the count per gap (s_waitcnt not counted) and the loop's instructions by mnemonic.
(tools/isa_mix.py gives the totals per loop; this gives the placement.  DESIGN 3w.)"""
import collections
import sys


def main():
  path, kern = sys.argv[1], sys.argv[2]
  lines = open(path).read().split('\n')
  start = [i for i, l in enumerate(lines) if l.startswith(kern)][0]
  end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
  body = lines[start:end]
  labels = {}
  for i, l in enumerate(body):
    if l.startswith('.LBB') and ':' in l:
      labels[l.split(':')[0]] = i
  best = None
  for i, l in enumerate(body):
    t = l.strip()
    if t.startswith('s_cbranch') or t.startswith('s_branch'):
      tgt = t.split()[-1]
      if tgt in labels and labels[tgt] < i:
        seg = [x.strip() for x in body[labels[tgt]:i + 1] if x.strip() and not x.strip().startswith(';') and not x.strip().startswith('.')]
        n = sum(1 for x in seg if x.startswith('v_mfma'))
        if n and (best is None or len(seg) < len(best)):
          best = seg
  gaps, cur = [], []
  for l in best:
    op = l.split()[0]
    if op.startswith('v_mfma'):
      gaps.append(cur)
      cur = []
    else:
      cur.append(op)
  gaps.append(cur)
  print('instructions in the loop: %d, MFMAs: %d' % (len(best), len(gaps) - 1))
  print('other instructions per gap:', [len([o for o in g if not o.startswith('s_waitcnt')]) for g in gaps])
  c = collections.Counter(o for g in gaps for o in g)
  print(c.most_common(40))


if __name__ == '__main__':
  main()
