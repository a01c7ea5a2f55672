#!/bin/bash
mkdir -p gpurun_out/flaky
for i in 1 2 3 4 5 6; do
  timeout 900 python -m pytest tests/test_gpu_data_parallel.py -x -q -k "rccl" 2>&1 | tail -2 | head -1
done
