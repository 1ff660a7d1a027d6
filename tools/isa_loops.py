#!/usr/bin/env python3
"""What the compiler made of the MFMA loops: reads the gfx950 assembly of the HIP sources (hipcc -S --cuda-device-only, one file per
source under --asm-dir, written by `--build`) and reports, per kernel, for its loop with the most MFMAs:

  scan      MFMAs, s_waitcnt vmcnt(0) among the vector-memory waits, loads / stores / scratch accesses / branches inside the loop
            (a vmcnt(0) inside a pipelined loop drains every request in flight: a spill reload, a request under a branch, a register
            reused as a load destination);
  timeline  where the requests and the waits sit, counted in MFMAs from the loop top (`40:Lx4` four loads after the 40th MFMA,
            `57:w32` an s_waitcnt vmcnt(32), `B` a barrier, `S` a store);
  bursts    runs of non-MFMA instructions between two MFMAs (vector, LDS, vector-memory, other) longer than 8: work the scheduler's
            group pattern had no slot for and left behind the MFMAs of a stage;
  chains    how many MFMAs lie between two writes of the same accumulator.

usage: python tools/isa_loops.py --build            (compile every csrc/*.hip to assembly, ~2 min)
       python tools/isa_loops.py scan [substring]
       python tools/isa_loops.py timeline|bursts|chains <substring of the mangled kernel name>
Round 5: profiles/r05_isa_loops_scan.txt, DESIGN.md 3r."""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'mode-2022_amd', 'csrc')
ASM = os.environ.get('MODE_ASM_DIR', '/tmp/mode_asm')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-munsafe-fp-atomics', '-I' + os.path.join(ROOT, 'include')]


def build():
  os.makedirs(ASM, exist_ok=True)
  procs = []
  for src in sorted(glob.glob(os.path.join(CSRC, '*.hip'))):
    out = os.path.join(ASM, os.path.basename(src)[:-4] + '.s')
    procs.append(subprocess.Popen(['/opt/rocm/bin/hipcc'] + FLAGS + ['-S', '--cuda-device-only', src, '-o', out], cwd=CSRC,
                                  stderr=subprocess.DEVNULL))
    while sum(p.poll() is None for p in procs) >= 6:
      procs[0].wait()
      procs = [p for p in procs if p.poll() is None]
  for p in procs:
    p.wait()
  print('assembly in', ASM)


def kernels(substr=''):
  for f in sorted(glob.glob(os.path.join(ASM, '*.s'))):
    s = open(f).read()
    for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end', s, re.S | re.M):
      if substr in m.group(1) and 'v_mfma' in m.group(2):
        yield os.path.basename(f)[:-2], m.group(1), m.group(2).split('\n')


def main_loop(body):
  """(first line, one past the last line) of the loop -- a backward branch to a label -- with the most MFMAs (the shortest such)."""
  labels = {}
  for n, l in enumerate(body):
    mm = re.match(r'^(\.LBB\d+_\d+):', l)
    if mm:
      labels[mm.group(1)] = n
  best = None
  for n, l in enumerate(body):
    mm = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
    if mm and mm.group(1) in labels and labels[mm.group(1)] < n:
      st = labels[mm.group(1)]
      mf = sum('v_mfma' in x for x in body[st:n + 1])
      if mf and (best is None or mf > best[0] or (mf == best[0] and n + 1 - st < best[2] - best[1])):
        best = (mf, st, n + 1)
  return best


def demangle(name):
  out = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
  return re.sub(r'\(anonymous namespace\)::', '', out).split('(')[0]


VMEM_LD = re.compile(r'(global|buffer|flat|scratch)_load')
VMEM_ST = re.compile(r'(global|buffer|flat|scratch)_store')


def scan(substr):
  for f, name, body in kernels(substr):
    lp = main_loop(body)
    if not lp:
      print('%-24s %-52s no loop' % (f, demangle(name)[:52]))
      continue
    seg = body[lp[1]:lp[2]]
    print('%-24s %-52s mfma %4d  vmcnt(0) %2d of %3d waits  loads %3d stores %3d scratch %2d branches %3d' %
          (f, demangle(name)[:52], lp[0], sum('vmcnt(0)' in x for x in seg), sum('vmcnt' in x for x in seg),
           sum(bool(VMEM_LD.search(x)) for x in seg), sum(bool(VMEM_ST.search(x)) for x in seg), sum('scratch_' in x for x in seg),
           sum('s_cbranch' in x for x in seg)))


def timeline(substr):
  for f, name, body in kernels(substr):
    lp = main_loop(body)
    if not lp:
      continue
    ev, nm = [], 0
    for l in body[lp[1]:lp[2]]:
      t = l.strip()
      if 'v_mfma' in t:
        nm += 1
      elif VMEM_LD.match(t):
        ev.append((nm, 'L'))
      elif VMEM_ST.match(t):
        ev.append((nm, 'S'))
      elif 'vmcnt' in t:
        ev.append((nm, 'w' + re.search(r'vmcnt\((\d+)\)', t).group(1)))
      elif 's_barrier' in t:
        ev.append((nm, 'B'))
    out, i = [], 0
    while i < len(ev):
      j = i
      while j < len(ev) and ev[j] == ev[i]:
        j += 1
      out.append('%d:%s%s' % (ev[i][0], ev[i][1], 'x%d' % (j - i) if j - i > 1 else ''))
      i = j
    print(demangle(name), 'mfma', lp[0])
    print('   ', ' '.join(out))


def bursts(substr):
  for f, name, body in kernels(substr):
    lp = main_loop(body)
    if not lp:
      continue
    gaps, cur = [], [0, 0, 0, 0]
    for l in body[lp[1]:lp[2]]:
      t = l.strip()
      if not t or t[0] in ';.':
        continue
      if 'v_mfma' in t:
        gaps.append(tuple(cur))
        cur = [0, 0, 0, 0]
      elif t.startswith('v_'):
        cur[0] += 1
      elif t.startswith('ds_'):
        cur[1] += 1
      elif re.match(r'(global|buffer|flat|scratch)_', t):
        cur[2] += 1
      else:
        cur[3] += 1
    gaps.append(tuple(cur))
    print(demangle(name), 'mfma', lp[0], 'vector', sum(g[0] for g in gaps), 'lds', sum(g[1] for g in gaps))
    print('    (vector, lds, vmem, other) in front of MFMA #i, where vector + lds + vmem > 8:',
          [(i, g) for i, g in enumerate(gaps) if g[0] + g[1] + g[2] > 8])


def chains(substr):
  for f, name, body in kernels(substr):
    lp = main_loop(body)
    if not lp:
      continue
    seq = []
    for l in body[lp[1]:lp[2]]:
      mm = re.match(r'v_mfma\S+\s+([av]\[\d+:\d+\])', l.strip())
      if mm:
        seq.append(mm.group(1))
    last, hist = {}, {}
    for i, a in enumerate(seq + seq[:8]):
      if a in last:
        hist[i - last[a]] = hist.get(i - last[a], 0) + 1
      last[a] = i
    print(demangle(name), 'mfma', len(seq), 'MFMAs between two writes of one accumulator -> count:', sorted(hist.items()))


if __name__ == '__main__':
  if len(sys.argv) > 1 and sys.argv[1] == '--build':
    build()
  elif len(sys.argv) > 1 and sys.argv[1] in ('scan', 'timeline', 'bursts', 'chains'):
    if not glob.glob(os.path.join(ASM, '*.s')):
      raise SystemExit('no assembly under %s: run with --build first' % ASM)
    {'scan': scan, 'timeline': timeline, 'bursts': bursts, 'chains': chains}[sys.argv[1]](sys.argv[2] if len(sys.argv) > 2 else '')
  else:
    raise SystemExit(__doc__)
