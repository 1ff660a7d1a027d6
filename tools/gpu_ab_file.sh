#!/bin/bash
# Same-box A/B of two versions of one source file: tools/gpu_ab_file.sh <file> <other-version> <timer command...>
# A = the tree as shipped, B = <other-version> copied over <file>; A again at the end.  The file is restored and the library rebuilt on
# ANY exit.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
F=$1; O=$2; shift 2
[ -f "$O" ] || { echo "gpu_ab_file.sh: no such file: $O" >&2; exit 2; }
ORIG=$(mktemp /tmp/ab_orig.XXXXXX)
cp $F $ORIG
restore() { if ! cmp -s $ORIG $F; then cp $ORIG $F; python mode-2022_amd/mode_hip/build.py 2>&1 | grep -v "^built\|up to date" | head -5; fi; rm -f $ORIG; }
trap restore EXIT
trap 'exit 130' INT TERM
echo "A : $($@ 2>&1 | tail -1)"
cp $O $F
if cmp -s $ORIG $F; then echo "gpu_ab_file.sh: $O is identical to $F" >&2; exit 2; fi
python mode-2022_amd/mode_hip/build.py 2>&1 | grep -v "^built\|up to date" | head -5
echo "B : $($@ 2>&1 | tail -1)"
cp $ORIG $F
python mode-2022_amd/mode_hip/build.py 2>&1 | grep -v "^built\|up to date" | head -5
echo "A': $($@ 2>&1 | tail -1)"
