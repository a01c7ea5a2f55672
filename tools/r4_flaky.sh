#!/bin/bash
mkdir -p gpurun_out/flaky
for i in 1 2 3 4; do
  timeout 900 python -m pytest tests/test_gpu_eve.py -x -q 2>&1 | grep -v Warning | tail -25 > gpurun_out/flaky/run$i.log
  tail -2 gpurun_out/flaky/run$i.log
done
