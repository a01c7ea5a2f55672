#!/bin/bash
mkdir -p gpurun_out/c3t
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_refinenet.py -x -q -k "eight_channel or refinenet" > gpurun_out/c3t/tests.log 2>&1
tail -4 gpurun_out/c3t/tests.log
python tools/refine_op_table.py 2>/dev/null | cut -c1-170 | grep "wgrad\|total" | head -24
python bench.py --workload c3 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('c3', d['value'], d['ms_per_step'])"
