#!/bin/bash
mkdir -p gpurun_out/misc
python tools/bench_wgrad_batch.py 1920 2>/dev/null | tail -3
python tools/bench_wgrad_batch.py 480 2>/dev/null | tail -1
timeout 900 python -m pytest tests/test_gpu_eyenet.py tests/test_gpu_kernels.py -x -q -k "one_node or trainer_takes or tail_chains or small_linear" 2>&1 | tail -2
for b in 32 8; do
python bench.py --steps 30 --warmup 10 --batch $b --no-c3 --no-c5 --no-points --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=$b', d['value'], d['ms_per_step'], d['roofline']['kernel'], round(d['roofline']['frac'],3), d['roofline']['avg_launch_ms'])"
done
python bench.py --workload c3 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c3', d['value'], d['ms_per_step'])"
