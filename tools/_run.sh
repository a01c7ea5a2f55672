python -m pytest tests/test_gpu_eve.py -m gpu -q -k configs4 2>&1 | grep -v "^$" | tail -25 > gpurun_out/t_c4.log
