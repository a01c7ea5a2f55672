cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_b8 -o b8 --output-format csv -- python $R/bench.py --batch 8 --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-c3 > $R/gpurun_out/b8.log 2>&1
cd $R
f=$(find gpurun_out/prof_b8 -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/b8_kernel_stats.csv
rm -rf gpurun_out/prof_b8
python bench.py --batch 8 --no-cpu-baseline --no-roofline --no-c3 > gpurun_out/b8.json 2>/dev/null
