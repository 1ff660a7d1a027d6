#!/bin/bash
# on the GPU box: SQ counters of conv3d_split_kernel (one rocprofv3 pass per counter group, kernel-trace only)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
L=$PWD/mode-2022_amd/mode_hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -Iinclude tools/experiments/conv3d_split_bench.cpp -L$L -lmode_hip -Wl,-rpath,$L -o /tmp/conv3d_split_bench || exit 1
OUT=$PWD/gpurun_out/pmc_split
mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_INSTS_SALU"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --pmc $grp --kernel-trace -d $OUT/g$i -o p --output-format csv -- /tmp/conv3d_split_bench > $OUT/g$i.log 2>&1)
  f=$(find $OUT/g$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
  k = r.get('Kernel_Name', '')
  if 'conv3d_split_kernel' in k or 'conv3d_kernel' in k:
    key = ('split' if 'split' in k else 'fp32 ', r['Counter_Name'])
    acc[key][0] += float(r['Counter_Value']); acc[key][1] += 1
for (k, c), (v, n) in sorted(acc.items()):
  print('%s %-28s %16.0f per launch (%d launches)' % (k, c, v / n, n))
PY
done
