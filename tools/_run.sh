timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_eyenet.py tests/test_gpu_bf16_parity.py tests/test_gpu_data_parallel.py -m gpu -q -x 2>&1 | grep -E "Error|error|assert|passed|failed" | head -20 > gpurun_out/t_all.log
python bench.py --no-cpu-baseline --no-c3 > gpurun_out/b_base.json 2> gpurun_out/b_base.err
python bench.py --no-cpu-baseline --no-c3 --batch 8 > gpurun_out/b_b8.json 2>> gpurun_out/b_base.err
