python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "conv" 2>&1 | grep -E "passed|failed|Error|assert" | head -20 > gpurun_out/t_conv.log
python bench.py --no-cpu-baseline --no-c3 > gpurun_out/bench_new.json 2>/dev/null
python tools/bench_conv.py > gpurun_out/bench_conv.txt 2>&1
