#!/bin/bash
# Memory-side counters of one operator at the benchmark shape (tools/one_kernel.py cases): address translation (UTCL1), vector-L1 and
# L2 request counts / latencies, texture-addresser busy.  Each group is its own --pmc pass; a group with a name this GPU does not have
# fails by itself.  usage: bash tools/pmc_mem.sh <tag> "<cases>"
TAG=${1:-pmcm}
CASES=${2:-"conv3d_bwd_weight_32"}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $OUT/avail.txt 2>&1
grep -o "TCP_[A-Z0-9_]*\|TA_[A-Z0-9_]*\|TCC_[A-Z0-9_]*\|SQ_[A-Z0-9_]*\|TD_[A-Z0-9_]*" $OUT/avail.txt | sort -u > $OUT/avail_names.txt
for C in $CASES; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${C}_trace -o t -- python3 $R/tools/one_kernel.py $C --no-flush > $OUT/${C}_trace.log 2>&1
  i=0
  for PASS in "GRBM_GUI_ACTIVE" \
              "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" \
              "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" \
              "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum" \
              "TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
              "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
              "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" \
              "TCC_TAG_STALL_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum" \
              "SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $PASS --output-format csv -d $OUT/${C}_p$i -o pmc -- python3 $R/tools/one_kernel.py $C > $OUT/${C}_p$i.log 2>&1 || echo "pass $i ($PASS) failed: $(tail -2 $OUT/${C}_p$i.log | cut -c1-200)"
  done
done
cd $R
python3 - <<PY > $OUT/summary.txt
import csv, glob, os, collections
out = '$OUT'
for case in '$CASES'.split():
  print('==', case)
  for f in glob.glob(os.path.join(out, case + '_trace', '**', '*kernel_stats.csv'), recursive=True):
    for row in list(csv.DictReader(open(f)))[:6]:
      print('  stats  %-70s calls %5s avg %10.1f us' % (row['Name'][:70], row['Calls'], float(row['AverageNs']) / 1e3))
  agg = collections.defaultdict(lambda: collections.defaultdict(float))
  cnt = collections.defaultdict(lambda: collections.defaultdict(int))
  for f in glob.glob(os.path.join(out, case + '_p*', '**', '*counter_collection.csv'), recursive=True):
    for row in csv.DictReader(open(f)):
      k = row['Kernel_Name'][:60]
      agg[k][row['Counter_Name']] += float(row['Counter_Value'])
      cnt[k][row['Counter_Name']] += 1
  for k in agg:
    if any(t in k for t in ('at::', 'elementwise', 'Fill', 'distribution')): continue
    print('  pmc   ', k)
    for c in sorted(agg[k]):
      print('      %-36s %16.0f per launch' % (c, agg[k][c] / max(cnt[k][c], 1)))
PY
cat $OUT/summary.txt | cut -c1-200
find $OUT -name "*.csv" -size +4M -delete
du -sh $OUT
