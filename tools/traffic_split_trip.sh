#!/bin/bash
# HBM traffic of the split-bf16 kernels (default arithmetic), as tools/traffic_trip.sh: one shape per run, FETCH_SIZE / WRITE_SIZE in
# separate --pmc passes -> gpurun_out/<tag>/traffic_raw.txt; then the SQ counters of the 3-D forward kernel (tools/experiments/pmc_split.sh).
TAG=${1:-traffic_split}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for CASE in conv3d_fwd_32 conv3d_bwd_data_32 conv3d_bwd_weight_32; do
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $C --output-format csv -d $OUT/$CASE/$C -o pmc -- python3 $R/tools/one_kernel.py $CASE > $OUT/${CASE}_$C.log 2>&1
    echo "$CASE $C rc=$?"
  done
  echo "== $CASE" >> $OUT/traffic_raw.txt
  python3 $R/tools/pmc_by_shape.py $OUT/$CASE >> $OUT/traffic_raw.txt 2>&1
done
find $OUT -name "*.csv" -size +4M -delete
grep -v "Fill\|fill\|vectorized\|elementwise\|reduce_kernel\|distribution\|rocclr" $OUT/traffic_raw.txt | cut -c1-160
cd $R && bash tools/experiments/pmc_split.sh > $OUT/pmc_sq.txt 2>&1; grep "split\|fp32" $OUT/pmc_sq.txt | cut -c1-120
