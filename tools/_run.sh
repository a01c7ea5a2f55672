python -m pytest tests/test_gpu_data_parallel.py -m gpu -q 2>&1 | grep -E "diverged|Error|passed|failed|assert" | head -30 > gpurun_out/t_dp.log
