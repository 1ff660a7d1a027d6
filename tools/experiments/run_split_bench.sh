#!/bin/bash
# on the GPU box: builds and runs tools/experiments/conv3d_split_bench.cpp against the in-tree libmode_hip.so
set -e
cd "$(dirname "$0")/../.."
L=$PWD/mode-2022_amd/mode_hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -Iinclude tools/experiments/conv3d_split_bench.cpp -L$L -lmode_hip -Wl,-rpath,$L -o /tmp/conv3d_split_bench
for a in "$@"; do /tmp/conv3d_split_bench $a || echo "exit $?"; done
[ $# -gt 0 ] || /tmp/conv3d_split_bench
