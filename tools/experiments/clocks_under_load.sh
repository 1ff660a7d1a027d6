# Engine clock and socket power while the training step replays (is the step power-limited?).  MODE_SPLIT_TALL as given.
cd $GRAFT_REPO_ROOT
env | grep -i "VISIBLE" 
python bench.py --steps 400 --warmup 5 --no-cpu-baseline --no-eval-b1 --no-collective-self-test --no-kernel-timing > /tmp/bench_load.log 2>&1 &
BP=$!
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket" | sed 's/.*GPU\[\([0-9]\)\].*: *\(.*\)/\1:\2/' | tr '\n' ' ' | cut -c1-600; echo
  sleep 1.5
done > /tmp/smi.log
grep '^{' /tmp/bench_load.log | cut -c1-200
tail -22 /tmp/smi.log
