#!/bin/bash
# quick check on the GPU box:  bash tools/r4_step.sh "<pytest -k expr>"  -> tests, then the configs[2] step
mkdir -p gpurun_out/r4step
timeout 1500 python -m pytest tests -m gpu -q -x -k "$1" 2>&1 | tail -4
python bench.py --workload c3 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('c3', d['value'], d['ms_per_step'])"
