python -m pytest tests/test_gpu_bf16_parity.py tests/test_gpu_kernels.py -m gpu -q 2>&1 | grep -v "^    \|^$" | tail -80 > gpurun_out/t1.log
