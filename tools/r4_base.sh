#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4base
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $O/prof -o b --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-c3 --no-points --no-roofline > $O/bench_profiled.log 2>&1
python $R/tools/dispatch_table.py $(find $O/prof -name "*kernel_trace.csv" | head -1) > $O/dispatch_b32.txt 2>&1
rm -rf $O/prof
timeout 600 rocprofv3 --kernel-trace -d $O/prof -o b --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --batch 8 --no-cpu-baseline --no-c3 --no-points --no-roofline > $O/bench_b8_profiled.log 2>&1
python $R/tools/dispatch_table.py $(find $O/prof -name "*kernel_trace.csv" | head -1) 50 > $O/dispatch_b8.txt 2>&1
rm -rf $O/prof
timeout 600 rocprofv3 --kernel-trace -d $O/prof -o b --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --batch 2 --seq 120 --size 256 --dtype fp16 --no-cpu-baseline --no-c3 --no-points --no-roofline > $O/bench_c5_profiled.log 2>&1
python $R/tools/dispatch_table.py $(find $O/prof -name "*kernel_trace.csv" | head -1) > $O/dispatch_c5.txt 2>&1
rm -rf $O/prof
timeout 600 rocprofv3 --kernel-trace -d $O/prof -o b --output-format csv -- python $R/tools/bench_eve.py --steps 2 > $O/c3_profiled.log 2>&1
python $R/tools/dispatch_table.py $(find $O/prof -name "*kernel_trace.csv" | head -1) 400 > $O/dispatch_c3.txt 2>&1
rm -rf $O/prof
cd $R
python tools/bench_in.py > $O/bench_in.txt 2>&1
python bench.py --no-cpu-baseline --no-c3 --no-points > $O/bench.json 2> $O/bench.err
