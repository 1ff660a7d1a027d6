#!/bin/bash
# Same-box A/B of one source edit: tools/gpu_ab.sh <file> <sed-expression> <timer command...>
# runs the timer on the tree as shipped (A), applies the sed expression, rebuilds, runs it again (B), restores, rebuilds, runs A again.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
F=$1; E=$2; shift 2
echo "A : $($@ 2>&1 | tail -1)"
cp $F /tmp/ab_orig
sed -i "$E" $F
python mode-2022_amd/mode_hip/build.py 2>&1 | grep -v "^built\|up to date" | head -5
echo "B : $($@ 2>&1 | tail -1)"
cp /tmp/ab_orig $F
python mode-2022_amd/mode_hip/build.py 2>&1 | grep -v "^built\|up to date" | head -5
echo "A': $($@ 2>&1 | tail -1)"
echo "B = A with: sed '$E' $F"
