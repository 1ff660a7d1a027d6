#!/bin/bash
# Register / LDS / scratch use of every kernel in one csrc/*.hip file (compile-only, no GPU needed).
# usage: bash tools/kernel_resources.sh mode-2022_amd/csrc/conv3d.hip [filter-regex]
R=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -I$R/include -I$R/mode-2022_amd/csrc \
  -Rpass-analysis=kernel-resource-usage -c "$1" -o /dev/null 2>&1 | python3 -c '
import re, subprocess, sys
cur, rows = None, []
for line in sys.stdin:
  m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", line)
  if not m: continue
  k, v = m.groups()
  if k == "Function Name":
    cur = {"name": v}; rows.append(cur)
  elif cur is not None:
    cur[k.split()[0] + ("Spill" if "Spill" in k else "")] = v
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
for r, n in zip(rows, names):
  n = re.sub(r"\(anonymous namespace\)::", "", n)
  print("%-90s vgpr %3s agpr %3s sgpr %3s scratch %4s occ %s spill s%s v%s lds %s" % (n[:90], r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize"), r.get("Occupancy"), r.get("SGPRsSpill"), r.get("VGPRsSpill"), r.get("LDS")))
' | grep -E "${2:-.}"
