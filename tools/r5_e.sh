#!/bin/bash
# round-5 fifth GPU call: in-graph join of the gated collectives, fast gate math in the 16-bit scan, ATen launch census
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5f
rm -rf $O; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_data_parallel.py tests/test_gpu_refinenet.py tests/test_gpu_bf16_parity.py tests/test_gpu_eve.py -m gpu -q -x --timeout 800 2>&1 | tail -15 > $O/pytest.log
tail -6 $O/pytest.log
Q="--no-cpu-baseline --no-c3 --no-c5 --no-points --no-roofline"
line() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$1', round(d['value']), 'frames/s', round(d['ms_per_step'],3), 'ms', 'gate_timeouts', d.get('gate_timeouts'))"; }
PORT=29817
for b in 8 32; do for mode in "--no-graph" "" "--graph-collectives"; do
  EVE_AMD_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$((PORT=PORT+1)) timeout 300 python bench.py --batch $b $mode $Q 2>>$O/err.log | line "rccl1 B=$b mode=[$mode]" >> $O/sweep.txt
done; done
python bench.py --batch 8 $Q 2>>$O/err.log | line "plain B=8" >> $O/sweep.txt
for w in c3 c5; do python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>>$O/err.log | line "$w" >> $O/sweep.txt; done
timeout 300 python tools/aten_ops_eve.py 8 > $O/aten_ops.txt 2>>$O/err.log
cat $O/sweep.txt; head -50 $O/aten_ops.txt; tail -3 $O/err.log
