#!/bin/bash
# full GPU suite + smoke + the round's measurement set on one box
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p $R/gpurun_out/r5final
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -30 > $R/gpurun_out/r5final/pytest_gpu.log
tail -4 $R/gpurun_out/r5final/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/r5_measure.sh
cp $R/gpurun_out/r5final/pytest_gpu.log $R/gpurun_out/final/profiles/r05_pytest_gpu.log
