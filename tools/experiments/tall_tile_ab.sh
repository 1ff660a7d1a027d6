# The 16-row tile of conv3d_split_kernel against the 8-row tile (MODE_SPLIT_TALL=0) inside the replayed step: rocprofv3 kernel stats
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in 1 0; do
  export MODE_SPLIT_TALL=$v
  OUT=$R/gpurun_out/tall$v; mkdir -p $OUT
  cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-eval-b1 > $OUT/rocprof.log 2>&1
  cd $R; for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -1); do python3 tools/profile_summary.py $f 40 > $OUT/summary.txt; done
  find $OUT/prof -name "*kernel_trace*.csv" -size +20M -delete
  echo "== MODE_SPLIT_TALL=$v"; grep '^{' $OUT/rocprof.log | cut -c1-120; head -24 $OUT/summary.txt | cut -c1-150
done
