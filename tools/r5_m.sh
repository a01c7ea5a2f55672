#!/bin/bash
# side-stream weight gradients: trainer tests + A/B at B = 8 / 16 / 32
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5m
rm -rf $O; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_eyenet.py tests/test_gpu_data_parallel.py tests/test_gpu_bf16_parity.py -m gpu -q -x --timeout 800 2>&1 | tail -8 > $O/pytest.log
tail -4 $O/pytest.log
Q="--no-cpu-baseline --no-c3 --no-c5 --no-points --no-roofline"
line() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$1', round(d['value']), 'frames/s', round(d['ms_per_step'],3), 'ms')"; }
for b in 8 16 32; do for m in 0 100000; do
  EVE_AMD_SIDE_WGRAD_MAX_IMAGES=$m python bench.py --batch $b $Q 2>>$O/err.log | line "B=$b side_wgrad_max_images=$m" >> $O/sweep.txt
done; done
EVE_AMD_SIDE_WGRAD_MAX_IMAGES=100000 python bench.py --batch 8 --no-graph $Q 2>>$O/err.log | line "B=8 eager side" >> $O/sweep.txt
PORT=30017
EVE_AMD_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT timeout 300 python bench.py --batch 8 $Q 2>>$O/err.log | line "rccl1 B=8 gated side" >> $O/sweep.txt
cat $O/sweep.txt; tail -3 $O/err.log
