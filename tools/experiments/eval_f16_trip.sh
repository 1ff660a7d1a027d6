cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_f16.py -m gpu -q -x --timeout 900 -s -k "eval_epilogues or eval_f16_layers" 2>&1 | grep -v "^$" | tail -60
python -m pytest tests/test_gpu_split.py tests/test_gpu_kernels.py -m gpu -q -x --timeout 900 -k "folded" 2>&1 | tail -5
python bench.py --mode eval --batch 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep "^{" | cut -c1-300
