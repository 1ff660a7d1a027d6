#!/bin/bash
# HBM traffic passes only (FETCH_SIZE, WRITE_SIZE in separate --pmc runs) for profiles/traffic.json:
#   3-D kernels at batch 2, spherical kernels at the step's 4 images per launch.
# usage: bash tools/traffic_trip.sh <tag>
TAG=${1:-traffic}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
for SET in "conv3d 2" "sphere 4"; do
  set -- $SET
  OUT=$R/gpurun_out/${TAG}_$1
  mkdir -p $OUT
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/$C -o pmc -- python3 $R/tools/microbench.py --only $1 --batch $2 --iters 2 > $OUT/$C.log 2>&1
    echo "$1 $C rc=$?"
  done
  python3 $R/tools/pmc_by_shape.py $OUT > $OUT/by_shape.txt 2>&1
  find $OUT -name "*.csv" -size +8M -delete
done
