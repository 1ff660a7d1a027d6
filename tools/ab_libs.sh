#!/bin/bash
# Same-box A B A B of library builds: runs a command once per variant and round with tools/experiments/libmode_hip_<variant>.so in
# the product library's place ("new" = the in-tree build as it was sent).  The product library is put back on ANY exit.
#   usage: bash tools/ab_libs.sh "<variants, e.g. r5 new>" <rounds> <command ...>
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
VARIANTS=$1; ROUNDS=$2; shift 2
LIB=mode-2022_amd/mode_hip/libmode_hip.so
cp $LIB /tmp/libmode_hip_product_$$.so
trap 'cp /tmp/libmode_hip_product_$$.so $LIB; rm -f /tmp/libmode_hip_product_$$.so' EXIT
trap 'exit 130' INT TERM
for r in $(seq 1 $ROUNDS); do
  for v in $VARIANTS; do
    if [ "$v" = new ]; then cp /tmp/libmode_hip_product_$$.so $LIB; else
      [ -f tools/experiments/libmode_hip_$v.so ] || { echo "tools/experiments/libmode_hip_$v.so is missing"; exit 2; }
      cp tools/experiments/libmode_hip_$v.so $LIB
    fi
    echo "== $v (round $r)"
    "$@" 2>&1 | tail -${AB_TAIL:-6}
  done
done
