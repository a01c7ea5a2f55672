#!/bin/bash
# round-5 second GPU call: the new scans + DP gates, then the rest of the GPU suite
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5b
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_refinenet.py tests/test_gpu_data_parallel.py -m gpu -q -x --timeout 600 2>&1 | tail -40 > $O/pytest_new.log
tail -15 $O/pytest_new.log
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 --deselect tests/test_gpu_refinenet.py --deselect tests/test_gpu_data_parallel.py 2>&1 | tail -30 > $O/pytest_rest.log
tail -8 $O/pytest_rest.log
