timeout 600 python -m pytest tests/test_gpu_refinenet.py tests/test_gpu_bf16_parity.py -m gpu -q -k "cgru or refinenet" 2>&1 | grep -E "passed|failed|Error|assert|rel" | head -30 > gpurun_out/t_cgru.log
timeout 600 python tools/bench_eve.py --steps 5 > gpurun_out/c3.log 2>&1
