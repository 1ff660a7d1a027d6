#!/bin/bash
# Same-box A/B of two versions of one source file: tools/gpu_ab_file.sh <file> <other-version> <timer command...>
# A = the tree as shipped, B = <other-version> copied over <file>; A again at the end.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
F=$1; O=$2; shift 2
echo "A : $($@ 2>&1 | tail -1)"
cp $F /tmp/ab_orig; cp $O $F
python mode-2022_amd/mode_hip/build.py 2>&1 | grep -v "^built\|up to date" | head -5
echo "B : $($@ 2>&1 | tail -1)"
cp /tmp/ab_orig $F
python mode-2022_amd/mode_hip/build.py 2>&1 | grep -v "^built\|up to date" | head -5
echo "A': $($@ 2>&1 | tail -1)"
