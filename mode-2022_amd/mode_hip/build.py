"""Build libmode_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python mode-2022_amd/mode_hip/build.py [--force]
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, 'csrc')
INCLUDE = os.path.join(ROOT, 'include')
OBJ = os.path.join(CSRC, 'obj')
LIB = os.path.join(HERE, 'libmode_hip.so')

HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-munsafe-fp-atomics', '-Wall', '-Wno-unused-function',
         '-I' + INCLUDE, '-I' + CSRC]
# debug builds of the experiments (e.g. MODE_HIP_DEFINES=MODE_TAPTIME for tools/experiments/sphere_taptime.py); never set by the product
FLAGS += ['-D' + d for d in os.environ.get('MODE_HIP_DEFINES', '').split()]


def sources():
  return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))


def _deps_mtime():
  hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
  hs += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE)]
  hs.append(os.path.abspath(__file__))
  return max(os.path.getmtime(h) for h in hs)


# Per-file flags.  -fno-slp-vectorize for the MFMA kernels whose staging arithmetic sits between the matrix instructions: the SLP
# vectoriser pairs their scalar fp32 operations into v_pk_*_f32, which do not overlap with MFMAs on gfx950, and as <2 x float> the
# fp16 split's remainder loses v_fma_mix_f32.
FILE_FLAGS = {
    'conv3d_split.hip': ['-fno-slp-vectorize'],
    'conv2d_split.hip': ['-fno-slp-vectorize'],
    'conv3d_split_wgrad.hip': ['-fno-slp-vectorize'],
    'conv2d_split_wgrad.hip': ['-fno-slp-vectorize'],
}


def _flags_of(src):
  return FLAGS + FILE_FLAGS.get(os.path.basename(src), [])


def _flags_hash(src):
  import hashlib
  return hashlib.sha256('\0'.join([HIPCC] + _flags_of(src)).encode()).hexdigest()


def _stamp(obj):
  return obj + '.flags'


def _compile(src, force):
  """Compile one source if its object is missing, older than the source / a header, or was built with other flags (the flags hash
  is stored PER OBJECT and written only after a successful compile: an interrupted build cannot leave a matching stamp beside
  objects of the old flags)."""
  obj = os.path.join(OBJ, os.path.basename(src)[:-4] + '.o')
  want = _flags_hash(src)
  try:
    with open(_stamp(obj)) as f:
      have = f.read().strip()
  except OSError:
    have = None
  if (not force and have == want and os.path.exists(obj) and
      os.path.getmtime(obj) > max(os.path.getmtime(src), _deps_mtime())):
    return obj, False
  cmd = [HIPCC] + _flags_of(src) + ['-c', src, '-o', obj]
  r = subprocess.run(cmd, capture_output=True, text=True)
  if r.returncode != 0:
    raise RuntimeError('hipcc failed: %s\n%s\n%s' % (' '.join(cmd), r.stdout, r.stderr))
  if r.stderr.strip():
    sys.stderr.write(r.stderr)
  with open(_stamp(obj), 'w') as f:
    f.write(want)
  return obj, True


def build(force=False, verbose=True):
  os.makedirs(OBJ, exist_ok=True)
  srcs = sources()
  with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
    results = list(ex.map(lambda s: _compile(s, force), srcs))
  objs = [o for o, _ in results]
  if force or any(c for _, c in results) or not os.path.exists(LIB):
    cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
      raise RuntimeError('link failed: %s\n%s\n%s' % (' '.join(cmd), r.stdout, r.stderr))
    if verbose:
      print('built', LIB)
  elif verbose:
    print('up to date', LIB)
  return LIB


if __name__ == '__main__':
  build(force='--force' in sys.argv)
