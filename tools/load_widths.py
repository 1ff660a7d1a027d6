#!/usr/bin/env python3
"""Global-memory access widths per kernel of libmode_hip.so, from its gfx950 ISA (needed to read FETCH_SIZE / WRITE_SIZE: the
counters tally requests, and the bytes per request differ by access width -- tools/calib/).

  python tools/load_widths.py [name filter]     -> per kernel: static counts of global_load/store _dword, _dwordx2, _dwordx3, _dwordx4
"""
import collections
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, 'mode-2022_amd', 'mode_hip', 'libmode_hip.so')
OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'


def main():
  flt = sys.argv[1] if len(sys.argv) > 1 else ''
  tmp = tempfile.mkdtemp()
  try:
    shutil.copy(LIB, tmp)
    subprocess.run([OBJDUMP, '--offloading', 'libmode_hip.so'], cwd=tmp, capture_output=True)
    for co in sorted(glob.glob(os.path.join(tmp, '*gfx950'))):
      asm = subprocess.run([OBJDUMP, '-d', co], capture_output=True, text=True).stdout
      cur, counts = None, collections.OrderedDict()
      for line in asm.splitlines():
        m = re.match(r'^[0-9a-f]+ <(.+)>:', line)
        if m:
          cur = m.group(1)
          counts[cur] = collections.Counter()
          continue
        m = re.search(r'\b(global_load|global_store|buffer_load|buffer_store)_(dword(?:x[234])?|ubyte|ushort|short|byte)\b', line)
        if m and cur:
          counts[cur][m.group(1).split('_')[1] + '_' + m.group(2)] += 1
      names = subprocess.run(['c++filt'], input='\n'.join(counts), capture_output=True, text=True).stdout.splitlines()
      for mangled, name in zip(counts, names):
        name = name.replace('(anonymous namespace)::', '').replace('void ', '')
        if counts[mangled] and flt in name:
          print('%-70s %s' % (name.split('(')[0][:70], '  '.join('%s=%d' % kv for kv in sorted(counts[mangled].items()))))
  finally:
    shutil.rmtree(tmp)


if __name__ == '__main__':
  main()
