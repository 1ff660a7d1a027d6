#!/bin/bash
# Same-box A/B of one source edit: tools/gpu_ab.sh <file> <sed-expression> <timer command...>
# runs the timer on the tree as shipped (A), applies the sed expression, rebuilds, runs it again (B), restores, rebuilds, runs A again.
# The edit is undone (file restored, library rebuilt) on ANY exit -- an interrupted run must not leave the tree or libmode_hip.so modified.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
F=$1; E=$2; shift 2
ORIG=$(mktemp /tmp/ab_orig.XXXXXX)
cp $F $ORIG
restore() { if ! cmp -s $ORIG $F; then cp $ORIG $F; python mode-2022_amd/mode_hip/build.py 2>&1 | grep -v "^built\|up to date" | head -5; fi; rm -f $ORIG; }
trap restore EXIT
trap 'exit 130' INT TERM
echo "A : $($@ 2>&1 | tail -1)"
sed -i "$E" $F
if cmp -s $ORIG $F; then echo "gpu_ab.sh: sed '$E' changed nothing in $F" >&2; exit 2; fi
python mode-2022_amd/mode_hip/build.py 2>&1 | grep -v "^built\|up to date" | head -5
echo "B : $($@ 2>&1 | tail -1)"
cp $ORIG $F
python mode-2022_amd/mode_hip/build.py 2>&1 | grep -v "^built\|up to date" | head -5
echo "A': $($@ 2>&1 | tail -1)"
echo "B = A with: sed '$E' $F"
