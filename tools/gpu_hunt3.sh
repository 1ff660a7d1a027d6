#!/bin/bash
# usage: bash tools/gpu_hunt3.sh <tag> <runs> <env assignments...>   -- hit counts of the two-rank hunt under debug variants
TAG=$1; RUNS=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
F='MIOpen\|^add \|^MODE\|^using\|amdgpu.ids'
for v in "$@"; do
  hits=0
  for rep in $(seq 1 $RUNS); do
    env $v timeout 600 python tools/determinism_hunt.py run --ranks 2 --steps 30 --replays 0 2>&1 | grep -v "$F" > $OUT/hunt_${v//[^A-Za-z0-9]/_}_$rep.log
    h=$(grep -c 'EAGER step' $OUT/hunt_${v//[^A-Za-z0-9]/_}_$rep.log)
    hits=$((hits + h))
    grep -h 'first at entry' $OUT/hunt_${v//[^A-Za-z0-9]/_}_$rep.log | sed 's/.*first at entry//' | cut -c1-90 | sort | uniq -c | head -5
  done
  echo "== $v: $hits differing steps in $RUNS runs x 2 ranks x 29 repeats"
done
