#!/bin/bash
# HBM traffic of the dominant kernels, one shape per run (tools/one_kernel.py, Infinity Cache flushed between launches), FETCH_SIZE
# and WRITE_SIZE in separate --pmc passes -> gpurun_out/<tag>/traffic_raw.txt (per kernel means, KiB).  profiles/traffic.json is
# made from it with the calibration of tools/calib (FETCH_SIZE x 2 for coalesced reads of any width, WRITE_SIZE x 1).
# usage: bash tools/traffic_trip.sh <tag>
TAG=${1:-traffic}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for CASE in conv3d_fwd_32 conv3d_bwd_data_32 conv3d_bwd_weight_32 sphere_fwd_t sphere_bwd_data_t sphere_bwd_weight_t cost_volume_fwd bn3d_32; do
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $C --output-format csv -d $OUT/$CASE/$C -o pmc -- python3 $R/tools/one_kernel.py $CASE > $OUT/${CASE}_$C.log 2>&1
    echo "$CASE $C rc=$?"
  done
  echo "== $CASE" >> $OUT/traffic_raw.txt
  python3 $R/tools/pmc_by_shape.py $OUT/$CASE >> $OUT/traffic_raw.txt 2>&1
done
find $OUT -name "*.csv" -size +4M -delete
grep -v "Fill\|fill\|vectorized\|elementwise\|reduce_kernel\|distribution\|rocclr" $OUT/traffic_raw.txt | cut -c1-140
