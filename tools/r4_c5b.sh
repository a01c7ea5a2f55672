#!/bin/bash
mkdir -p gpurun_out/c5b
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_eyenet.py -x -q -k "stem_wgrad or 256x256" > gpurun_out/c5b/tests.log 2>&1
tail -3 gpurun_out/c5b/tests.log
python bench.py --workload c5 --steps 4 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c5', d['value'], d['ms_per_step'])"
