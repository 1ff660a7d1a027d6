#!/bin/bash
# PMC counter passes (separate from kernel-trace/stats, as the guide prescribes) on the micro-benchmarks.
# usage: bash tools/pmc_trip.sh <tag> <microbench --only list>
TAG=${1:-pmc}
ONLY=${2:-sphere,conv3d}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
for PASS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
            "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
            "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" \
            "FETCH_SIZE" "WRITE_SIZE"; do
  N=$(echo $PASS | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --pmc $PASS --output-format csv -d $OUT/$N -o pmc -- python3 $R/tools/microbench.py --only $ONLY --iters 2 $MB_EXTRA > $OUT/$N.log 2>&1
  echo "pass $N rc=$?"
done
cd $R
python3 tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt | cut -c1-250
find $OUT -name "*.csv" -size +8M -delete
du -sh $OUT
