#!/bin/bash
mkdir -p gpurun_out/c3s
timeout 1200 python -m pytest tests/test_gpu_kernels.py -x -q -k "row_streaming or narrow_output or conv_fwd_dgrad" > gpurun_out/c3s/tests.log 2>&1
tail -4 gpurun_out/c3s/tests.log
timeout 1200 python -m pytest tests/test_gpu_refinenet.py -x -q > gpurun_out/c3s/refine.log 2>&1
tail -2 gpurun_out/c3s/refine.log
python tools/refine_op_table.py 2>/dev/null | cut -c1-170 | grep "conv2d\|total" | head -40 > gpurun_out/c3s/ops.txt
head -30 gpurun_out/c3s/ops.txt
for v in 1 0; do
EVE_CONV3X3_STREAM=$v python bench.py --workload c3 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('c3 stream3x3=$v', d['value'], d['ms_per_step'])"
done
