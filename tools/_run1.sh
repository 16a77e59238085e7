python -m pytest tests/test_gpu_parity.py tests/test_gpu_bf16.py tests/test_gpu_modules.py tests/test_00_gpu_two_ranks.py -q -m gpu -x 2>&1 | grep -E "^E  |^FAILED|passed|failed" | cut -c1-300 | head
bash tools/run_timeline2.sh cfg5s4 --config cfg5
