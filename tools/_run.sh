timeout 900 python -m pytest tests/test_gpu_data_parallel.py -m gpu -q -x -k rccl 2>&1 | grep -E "Error|error|assert|passed|failed" | head -30 > gpurun_out/t_all.log
