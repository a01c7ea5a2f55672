#!/bin/bash
mkdir -p gpurun_out/b8p
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_b8 -o b --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --batch 8 --steps 20 --warmup 2 --no-cpu-baseline --no-c3 --no-c5 --no-points --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/b8p/log.txt 2>&1
cp $(find /tmp/prof_b8 -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/b8p/b8_kernel_stats.csv
grep '^{' $GRAFT_REPO_ROOT/gpurun_out/b8p/log.txt | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"
