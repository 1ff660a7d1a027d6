#!/bin/bash
# two concurrent copies of each variant of tools/experiments/adj9_race.py;  usage: bash tools/experiments/adj9_race.sh <tag>
TAG=${1:-race}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
F='MIOpen\|amdgpu.ids'
pair() {  # name, args...
  local name=$1; shift
  python tools/experiments/adj9_race.py --tag ${name}-A "$@" > $OUT/$name.A.log 2>&1 &
  local pa=$!
  python tools/experiments/adj9_race.py --tag ${name}-B "$@" > $OUT/$name.B.log 2>&1 &
  local pb=$!
  wait $pa $pb
  grep -h "differ" $OUT/$name.A.log $OUT/$name.B.log | grep -v "$F" | cut -c1-330 | tail -8
}
echo "== single process, chain"; python tools/experiments/adj9_race.py --tag single --seconds 20 2>&1 | grep differ | cut -c1-330 | tail -4
echo "== two processes, chain"; pair chain --seconds 40
echo "== two processes, chain + sync between kernels"; pair chainsync --sync --seconds 40 --iters 500
echo "== two processes, adjoint kernel only (fixed gyt)"; pair adj --only adj --seconds 40
echo "== two processes, transposes only"; pair tr --only transpose --seconds 30
echo "== two processes, chain at 128x64"; pair chain128 --H 128 --W 64 --seconds 40 --iters 500
