#!/bin/bash
# round-5 first GPU call: GPU tests of the groundwork tree, the bench line, the trained-network 16-bit cost
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5a
rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -30 > $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
timeout 900 python tools/train_sanity.py 400 bf16 > $O/train_sanity.log 2>&1
tail -5 $O/pytest_gpu.log; cut -c1-1500 $O/bench.json; tail -12 $O/train_sanity.log
