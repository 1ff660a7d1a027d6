#!/bin/bash
# The instruction-level experiments behind DESIGN.md 6.0 (round 4), one file each under gpurun_out/<tag>/:
#   bash tools/experiments/run_microbench.sh <tag>
TAG=${1:-mb}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
B="hipcc --offload-arch=gfx950 -O3"
$B -o /tmp/mvo tools/experiments/mfma_valu_overlap.hip 2>/dev/null && /tmp/mvo > $OUT/mfma_valu_overlap.txt
$B -o /tmp/moc tools/experiments/mfma_op_cost.hip 2>/dev/null && /tmp/moc > $OUT/mfma_op_cost.txt
$B -o /tmp/ldsr tools/experiments/lds_rate.hip 2>/dev/null && /tmp/ldsr > $OUT/lds_rate.txt
$B -o /tmp/tap tools/experiments/tap_pipeline.hip 2>/dev/null && /tmp/tap > $OUT/tap_pipeline.txt
python3 tools/experiments/gen_tap_asm.py > /tmp/tap_asm.hip && $B -o /tmp/tap_asm /tmp/tap_asm.hip 2>/dev/null && /tmp/tap_asm > $OUT/tap_asm.txt
python3 tools/experiments/sphere_small_tiles_only.py 2>/dev/null | grep images > $OUT/sphere_fwd_small_tiles_only.txt
wc -l $OUT/*.txt
