#!/bin/bash
# quick GPU validation through gpurun:  bash tools/gpu_check.sh [pytest args...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/check
rm -rf $O; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q -x "$@" 2>&1 | tail -25 > $O/pytest_gpu.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-c3 > $O/bench_bf16.json 2> $O/bench.err
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-c3 --dtype fp16 > $O/bench_fp16.json 2>> $O/bench.err
tail -5 $O/pytest_gpu.log; cat $O/bench_bf16.json | cut -c1-400; cat $O/bench_fp16.json | cut -c1-400; tail -5 $O/bench.err
