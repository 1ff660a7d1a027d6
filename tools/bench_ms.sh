#!/bin/bash
# one number: ms per training step of the default bench (no baseline / eval / per-kernel legs), for same-box A/B runs
python bench.py --steps ${1:-10} --warmup 3 --no-cpu-baseline --no-eval-b1 --no-kernel-timing 2>/dev/null | grep '^{' | python -c "import json,sys; print('%.3f ms per step' % json.loads(sys.stdin.read())['ms_per_step'])"
