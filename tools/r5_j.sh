#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5l
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_eyenet.py -m gpu -q -x --timeout 600 -k "stem or trunk or configs or independent or instnorm" 2>&1 | tail -6 > $O/pytest.log
tail -4 $O/pytest.log
timeout 900 python -m pytest tests/test_gpu_eve.py -m gpu -q --timeout 800 -k "configs4" 2>&1 | tail -4 >> $O/pytest.log
tail -3 $O/pytest.log
line() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$1', round(d['value']), 'frames/s', round(d['ms_per_step'],3), 'ms')"; }
for w in c3 c5; do python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>>$O/err.log | line "$w" >> $O/sweep.txt; done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_c5 -o c --output-format csv -- python $R/bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/c5_profiled.log 2>&1
cp $(find $O/prof_c5 -name "*kernel_stats.csv" | head -1) $O/c5_kernel_stats.csv; rm -rf $O/prof_c5
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_c3 -o c --output-format csv -- python $R/tools/bench_eve.py --steps 5 > $O/c3_profiled.log 2>&1
cp $(find $O/prof_c3 -name "*kernel_stats.csv" | head -1) $O/c3_kernel_stats.csv; rm -rf $O/prof_c3
cd $R; cat $O/sweep.txt; tail -3 $O/err.log
