#!/bin/bash
# GPU session for the determinism hunt.  usage: bash tools/gpu_hunt.sh <tag> [pairs]
TAG=${1:-hunt}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
F='MIOpen\|^add \|^MODE\|^using\|amdgpu.ids'
echo "== hunt: 2 ranks, tiny, clones kept"; timeout 900 python tools/determinism_hunt.py run --ranks 2 --steps 25 --replays 25 --keep 2>&1 | grep -v "$F" > $OUT/hunt_2r_keep.log; grep -c . $OUT/hunt_2r_keep.log; grep 'differ\|NaN\|Error\|error' $OUT/hunt_2r_keep.log | cut -c1-600 | head -40
echo "== hunt: 2 ranks, tiny, NaN-filled allocations"; timeout 900 python tools/determinism_hunt.py run --ranks 2 --steps 10 --replays 10 --keep --nanfill 2>&1 | grep -v "$F" > $OUT/hunt_2r_nan.log; grep 'differ\|NaN\|Error\|error' $OUT/hunt_2r_nan.log | cut -c1-600 | head -40
echo "== hunt: 1 rank + hog, tiny"; timeout 900 python tools/determinism_hunt.py run --ranks 1 --steps 25 --replays 25 --keep --hog 2>&1 | grep -v "$F" > $OUT/hunt_1r_hog.log; grep 'differ\|NaN\|Error\|error' $OUT/hunt_1r_hog.log | cut -c1-600 | head -40
echo "== hunt: 2 ranks, 512x256/64, checksums"; timeout 900 python tools/determinism_hunt.py run --ranks 2 --steps 10 --replays 10 --maxdisp 64 --H 512 --W 256 2>&1 | grep -v "$F" > $OUT/hunt_2r_cfg1.log; grep 'differ\|NaN\|Error\|error' $OUT/hunt_2r_cfg1.log | cut -c1-600 | head -40
echo "== pair loop"; timeout 1500 python tools/two_rank_pair_loop.py ${2:-8} 2>&1 | grep -v "$F" > $OUT/pair_loop.log; grep 'differ\|identical' $OUT/pair_loop.log | cut -c1-400 | tail -30
