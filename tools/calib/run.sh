#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration per access width (tools/calib/traffic_calib.hip) -> gpurun_out/<tag>/calibration.txt
# usage (on the GPU box, from the repo root): bash tools/calib/run.sh <tag> [MiB]
TAG=${1:-calib}
MIB=${2:-2048}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/traffic_calib $R/tools/calib/traffic_calib.hip || exit 1
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $OUT/$C -o pmc -- /tmp/traffic_calib $MIB > $OUT/$C.log 2>&1
  echo "$C rc=$?"
done
python3 - "$OUT" "$MIB" > $OUT/calibration.txt <<'PY'
import collections, csv, glob, os, sys
root, mib = sys.argv[1], int(sys.argv[2])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, '*', '*counter_collection.csv')):
  for r in csv.DictReader(open(f)):
    agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
print('known bytes per kernel launch: %d MiB = %d KiB' % (mib, mib * 1024))
for name in sorted(agg):
  short = name.replace('void ', '').split('(')[0]
  for c, v in sorted(agg[name].items()):
    m = sum(v) / len(v)
    print('%-48s %-10s n=%d mean %14.1f KiB   counter / known = %.4f' % (short, c, len(v), m, m / (mib * 1024)))
PY
cat $OUT/calibration.txt
find $OUT -name "*.csv" -size +4M -delete
