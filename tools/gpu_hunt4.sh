#!/bin/bash
# usage: bash tools/gpu_hunt4.sh <tag> <tiny runs> <full runs>  -- hit counts of the two-rank hunt (eager repeats + graph replays) at two sizes
TAG=$1; NT=$2; NF=$3
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
F='MIOpen\|^add \|^MODE\|^using\|amdgpu.ids'
for rep in $(seq 1 $NT); do
  timeout 600 python tools/determinism_hunt.py run --ranks 2 --steps 30 --replays 10 2>&1 | grep -v "$F" > $OUT/tiny_$rep.log
  grep -h 'first at entry\|GRAPH' $OUT/tiny_$rep.log | cut -c1-200 | head -6
done
echo "== tiny: $(cat $OUT/tiny_*.log | grep -c 'EAGER step') differing eager steps, $(cat $OUT/tiny_*.log | grep -c 'GRAPH replay') differing replays in $NT runs x 2 ranks x (29 + 10)"
for rep in $(seq 1 $NF); do
  timeout 1000 python tools/determinism_hunt.py run --ranks 2 --steps 15 --replays 10 --maxdisp 192 --H 1024 --W 512 2>&1 | grep -v "$F" > $OUT/full_$rep.log
  grep -h 'first at entry\|GRAPH' $OUT/full_$rep.log | cut -c1-260 | head -6
done
echo "== full: $(cat $OUT/full_*.log | grep -c 'EAGER step') differing eager steps, $(cat $OUT/full_*.log | grep -c 'GRAPH replay') differing replays in $NF runs x 2 ranks x (14 + 10)"
