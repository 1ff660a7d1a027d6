#!/bin/bash
# Where does the stride-2 weight-gradient split kernel spend its time?  As shipped, without MFMAs, without global loads (same box).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
F=mode-2022_amd/csrc/conv3d_split_wgrad_s2.hip
for A in 0 1 2; do
  sed -i "s/^#define MODE_S2W_ABLATE [0-9]/#define MODE_S2W_ABLATE $A/" $F
  python mode-2022_amd/mode_hip/build.py 2>&1 | grep -v "^built\|up to date" | head -5
  echo "ablate $A: $(python tools/time_c3d.py s2w 2>&1 | tail -1)"
done
sed -i "s/^#define MODE_S2W_ABLATE [0-9]/#define MODE_S2W_ABLATE 0/" $F
