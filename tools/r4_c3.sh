#!/bin/bash
# C3 work: kernel tests of the new paths, RefineNet parity, the op table, the C3 step
mkdir -p gpurun_out/c3w
timeout 1200 python -m pytest tests/test_gpu_kernels.py -x -q -k "streaming_1x1 or narrow_output or accumulates or conv_fwd_dgrad or instnorm" > gpurun_out/c3w/kernels.log 2>&1
tail -4 gpurun_out/c3w/kernels.log
timeout 1200 python -m pytest tests/test_gpu_refinenet.py tests/test_gpu_bf16_parity.py -x -q > gpurun_out/c3w/refine.log 2>&1
tail -4 gpurun_out/c3w/refine.log
python tools/refine_op_table.py 2>/dev/null | cut -c1-175 > gpurun_out/c3w/ops.txt
head -3 gpurun_out/c3w/ops.txt
python bench.py --workload c3 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c3', d['value'], d['ms_per_step'])"
EVE_IN_BIG_PLANES=0 python bench.py --workload c3 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c3 big=0', d['value'], d['ms_per_step'])"
EVE_CONV1X1_STREAM=0 python bench.py --workload c3 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c3 stream=0', d['value'], d['ms_per_step'])"
