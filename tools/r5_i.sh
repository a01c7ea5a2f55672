#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5i
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_refinenet.py -m gpu -q -x --timeout 600 2>&1 | tail -15 > $O/pytest_scan.log
tail -6 $O/pytest_scan.log
timeout 900 python -m pytest tests/test_gpu_bf16_parity.py tests/test_gpu_eve.py -m gpu -q --timeout 800 2>&1 | tail -8 > $O/pytest_rest.log
tail -4 $O/pytest_rest.log
line() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$1', round(d['value']), 'frames/s', round(d['ms_per_step'],3), 'ms')"; }
for w in c3 c5; do python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>>$O/err.log | line "$w" >> $O/sweep.txt; done
for w in c3 c5; do EVE_CGRU_SEQ_MAX_B=0 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>>$O/err.log | line "$w three-per-workgroup" >> $O/sweep.txt; done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_c5 -o c --output-format csv -- python $R/bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/c5_profiled.log 2>&1
grep "cgru_scan" $(find $O/prof_c5 -name "*kernel_stats.csv" | head -1) | cut -d, -f1-4 | cut -c1-60,150-260 > $O/c5_scan.txt; rm -rf $O/prof_c5
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_c3 -o c --output-format csv -- python $R/tools/bench_eve.py --steps 5 > $O/c3_profiled.log 2>&1
grep "cgru_scan" $(find $O/prof_c3 -name "*kernel_stats.csv" | head -1) | cut -d, -f1-4 | cut -c1-60,150-260 > $O/c3_scan.txt; rm -rf $O/prof_c3
cd $R; cat $O/sweep.txt $O/c5_scan.txt $O/c3_scan.txt; tail -3 $O/err.log
