#!/bin/bash
# usage: bash tools/gpu_hunt2.sh <tag>   -- the hunt with tensor dumps of the first differing operator call + the repaired graph tests
TAG=${1:-hunt2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
F='MIOpen\|^add \|^MODE\|^using\|amdgpu.ids'
for rep in 1 2 3; do
  echo "== hunt $rep: 2 ranks, tiny, clones + inputs kept"
  timeout 900 python tools/determinism_hunt.py run --ranks 2 --steps 30 --replays 5 --keep --dump $OUT/dump$rep 2>&1 | grep -v "$F" > $OUT/hunt$rep.log
  grep 'first at entry\|eager:\|graph:\|dumped\|exit code' $OUT/hunt$rep.log | cut -c1-500 | head -20
done
echo "(tests skipped)"
