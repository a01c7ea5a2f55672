#!/bin/bash
mkdir -p gpurun_out/in2
timeout 1200 python -m pytest tests/test_gpu_kernels.py -x -q -k "instnorm" > gpurun_out/in2/tests.log 2>&1
tail -3 gpurun_out/in2/tests.log
python tools/bench_in_refine.py 2>/dev/null | grep "big=0"
python tools/refine_op_table.py 2>/dev/null | cut -c1-150 | grep "instnorm\|total" | head -30
python bench.py --workload c3 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c3', d['value'], d['ms_per_step'])"
