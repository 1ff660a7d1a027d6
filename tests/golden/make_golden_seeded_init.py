#!/usr/bin/env python3
"""tests/golden/seeded_init.json: SHA-256 of the state_dict the IMPORTED reference constructs under torch.manual_seed(123), per
model -- seeded from-scratch runs of the drop-in modules must start from bit-identical weights (same RNG draws in the same order,
including the reference's unregistered `downsample` projections).  Development container only (needs /root/reference).
Usage: python tests/golden/make_golden_seeded_init.py"""
import hashlib
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

CASES = {
    'ModeFusion': ('ModeFusion', (1000, [32, 64, 128, 256], {'depth': 12, 'rgb': 12})),
    'Baseline': ('Baseline', (1000,)),
    'ModeDisparity_Sphere': ('ModeDisparity', (64, 'Sphere', 128, 64, 'Cassini')),
    'ModeDisparity_Regular': ('ModeDisparity', (64, 'Regular')),
}


def digest(sd):
  h = hashlib.sha256()
  for k, v in sd.items():
    h.update(k.encode())
    h.update(v.detach().cpu().contiguous().numpy().tobytes())
  return h.hexdigest()


def main():
  models, _ = mg.import_reference()
  out = {'seed': 123}
  for tag, (cls, args) in CASES.items():
    torch.manual_seed(123)
    out[tag] = digest(getattr(models, cls)(*args).state_dict())
    print(tag, out[tag][:16])
  with open(os.path.join(HERE, 'seeded_init.json'), 'w') as f:
    json.dump(out, f, indent=1)


if __name__ == '__main__':
  main()
