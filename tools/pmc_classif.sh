#!/bin/bash
# PMC counter passes (separate from kernel-trace/stats, as the guide prescribes) on tools/experiments/classif_bench.py.
TAG=${1:-pmc_classif}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for PASS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
            "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
            "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" \
            "FETCH_SIZE" "WRITE_SIZE"; do
  N=$(echo $PASS | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --pmc $PASS --output-format csv -d $OUT/$N -o pmc -- python3 $R/tools/experiments/classif_bench.py fused > $OUT/$N.log 2>&1
  echo "pass $N rc=$?"
done
cd $R
python3 - $OUT <<'P' | tee $OUT/summary.txt
import collections, csv, glob, os, sys
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
  for r in csv.DictReader(open(f)):
    name = r.get('Kernel_Name', '').replace('(anonymous namespace)::', '').replace('void ', '')[:40]
    agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
for name in sorted(agg):
  if not any(k in name for k in ('classif', 'bn_stats', 'head')):
    continue
  a = {c: sum(v) / len(v) for c, v in agg[name].items()}
  print(name)
  for c in sorted(a):
    print('    %-28s %14.1f' % (c, a[c]))
  if a.get('SQ_INSTS_MFMA'):
    print('    -> VALU per MFMA %.2f, matrix pipe busy %.1f %% of GRBM_GUI_ACTIVE x 1024 SIMDs / 8, wave cycles in s_waitcnt %.1f %%' % (
        a['SQ_INSTS_VALU'] / a['SQ_INSTS_MFMA'], 100 * a['SQ_VALU_MFMA_BUSY_CYCLES'] / (a.get('GRBM_GUI_ACTIVE', 0) / 8 * 1024 + 1e-9),
        100 * a['SQ_WAIT_ANY'] / a['SQ_WAVE_CYCLES']))
  if 'FETCH_SIZE' in a:
    print('    -> HBM traffic 2 x FETCH + WRITE = %.1f MB' % ((2 * a['FETCH_SIZE'] + a.get('WRITE_SIZE', 0)) * 1024 / 1e6))
P
find $OUT -name "*.csv" -size +8M -delete
