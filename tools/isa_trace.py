#!/usr/bin/env python3
"""Instruction-class trace of one kernel in a gfx950 assembly listing (hipcc -S --cuda-device-only): M = MFMA, r / w = LDS read / write,
G = global load, S = global store, v / s = other vector / scalar (runs compressed), [..] = s_waitcnt, B = barrier, | = label.
usage: isa_trace.py file.s kernel-name-substring"""
import re
import sys

lines = open(sys.argv[1]).read().splitlines()
start = [i for i, l in enumerate(lines) if re.match(r'^_Z\S*%s\S*:' % re.escape(sys.argv[2]), l)][0]
seq = []
for l in lines[start + 1:]:
  t = l.strip().split()
  if not t or t[0].startswith((';', '//')):
    continue
  op = t[0]
  if op.endswith(':'):
    seq.append('\n|%s ' % op)
    continue
  if op.startswith('.'):
    continue
  if op.startswith('v_mfma'):
    c = 'M'
  elif op.startswith(('ds_read', 'ds_load')):
    c = 'r'
  elif op.startswith(('ds_write', 'ds_store')):
    c = 'w'
  elif op.startswith(('global_load', 'buffer_load')):
    c = 'G'
  elif op.startswith(('global_store', 'buffer_store')):
    c = 'S'
  elif op.startswith('s_waitcnt'):
    c = '[%s]' % ''.join(t[1:])
  elif op.startswith('s_barrier'):
    c = 'B'
  elif op.startswith(('s_cbranch', 's_branch')):
    c = 'J'
  elif op.startswith('v_'):
    c = 'v'
  elif op.startswith('s_'):
    c = 's'
  else:
    c = '?'
  seq.append(c)
  if op == 's_endpgm':
    break
out = ''.join(seq)
out = re.sub(r'v{3,}', lambda m: 'v%d.' % len(m.group(0)), out)
out = re.sub(r's{3,}', lambda m: 's%d.' % len(m.group(0)), out)
print(out)
