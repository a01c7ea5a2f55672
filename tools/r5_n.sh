#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5n
rm -rf $O; mkdir -p $O
cd $R
Q="--no-cpu-baseline --no-c3 --no-c5 --no-points --no-roofline"
line() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$1', round(d['value']), 'frames/s', round(d['ms_per_step'],3), 'ms')"; }
for b in 8 16 32; do for m in 0 100000; do
  EVE_AMD_SIDE_WGRAD_MAX_IMAGES=$m python bench.py --batch $b $Q 2>>$O/err.log | line "B=$b side_wgrad_max_images=$m" >> $O/sweep.txt
done; done
python bench.py --batch 8 --no-graph $Q 2>>$O/err.log | line "B=8 eager" >> $O/sweep.txt
EVE_AMD_SIDE_WGRAD_MAX_IMAGES=100000 python bench.py --batch 8 --no-graph $Q 2>>$O/err.log | line "B=8 eager side" >> $O/sweep.txt
EVE_AMD_SIDE_WGRAD_MAX_IMAGES=100000 timeout 900 python -m pytest tests/test_gpu_eyenet.py tests/test_gpu_data_parallel.py -m gpu -q -x --timeout 800 2>&1 | tail -3 > $O/pytest.log
cat $O/sweep.txt $O/pytest.log; tail -3 $O/err.log
