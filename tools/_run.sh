timeout 1500 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "linear" 2>&1 | grep -E "Error|error|assert|passed|failed" | head -20 > gpurun_out/t_all.log
python bench.py --no-cpu-baseline --no-c3 > gpurun_out/b_base.json 2> gpurun_out/b_base.err
python bench.py --no-cpu-baseline --no-c3 --batch 8 > gpurun_out/b_b8.json 2>> gpurun_out/b_base.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_x -o b --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-c3 --no-roofline > /dev/null 2>&1
cp $(find $GRAFT_REPO_ROOT/gpurun_out/prof_x -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/x_kernel_stats.csv; rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_x
