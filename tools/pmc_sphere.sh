#!/bin/bash
# Kernel durations + SQ counters + HBM traffic of the spherical operators at the benchmark shape (128 -> 128, 256 x 128, 4 images,
# plane-transposed storage): separate passes, as the guide prescribes.  usage: bash tools/pmc_sphere.sh <tag> [cases]
TAG=${1:-pmcs}
CASES=${2:-"sphere_bwd_weight_t sphere_bwd_data_t"}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for C in $CASES; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${C}_trace -o t -- python3 $R/tools/one_kernel.py $C --no-flush > $OUT/${C}_trace.log 2>&1
  for PASS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
              "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
              "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
    N=$(echo $PASS | tr ' ' '_' | cut -c1-30)
    timeout 300 rocprofv3 --pmc $PASS --output-format csv -d $OUT/${C}_$N -o pmc -- python3 $R/tools/one_kernel.py $C > $OUT/${C}_$N.log 2>&1
  done
done
cd $R
python3 - <<PY > $OUT/summary.txt
import csv, glob, os, collections
out = '$OUT'
for case in '$CASES'.split():
  print('==', case)
  for f in glob.glob(os.path.join(out, case + '_trace', '**', '*kernel_stats.csv'), recursive=True):
    for row in list(csv.DictReader(open(f)))[:8]:
      print('  stats  %-70s calls %5s avg %10.1f us' % (row['Name'][:70], row['Calls'], float(row['AverageNs']) / 1e3))
  agg = collections.defaultdict(lambda: collections.defaultdict(float))
  cnt = collections.defaultdict(lambda: collections.defaultdict(int))
  for f in glob.glob(os.path.join(out, case + '_*', '**', '*counter_collection.csv'), recursive=True):
    for row in csv.DictReader(open(f)):
      k = row['Kernel_Name'][:60]
      agg[k][row['Counter_Name']] += float(row['Counter_Value'])
      cnt[k][row['Counter_Name']] += 1
  for k in agg:
    if not any(t in k for t in ('sphere', 'reduce_gw', 'conv3d', 'deconv3d')): continue
    print('  pmc   ', k)
    for c in sorted(agg[k]):
      print('      %-28s %16.0f per launch' % (c, agg[k][c] / max(cnt[k][c], 1)))
PY
cat $OUT/summary.txt | cut -c1-200
find $OUT -name "*.csv" -size +4M -delete
du -sh $OUT
