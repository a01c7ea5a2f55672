cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c3 -- python $R/tools/bench_eve.py --steps 5 > $R/gpurun_out/c3.log 2>&1
cd $R
python tools/kstats.py $(ls gpurun_out/prof_c3/*/*kernel_stats.csv | head -1) 7 0.3 > gpurun_out/c3_kstats.txt
