#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5h
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_data_parallel.py tests/test_gpu_bf16_parity.py -m gpu -q --timeout 800 2>&1 | tail -8 > $O/pytest.log
tail -4 $O/pytest.log
Q="--no-cpu-baseline --no-c3 --no-c5 --no-points --no-roofline"
line() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$1', round(d['value']), 'frames/s', round(d['ms_per_step'],3), 'ms', 'gate_timeouts', d.get('gate_timeouts'))"; }
PORT=29917
for b in 8 32; do
for var in "EVE_AMD_COMM_PRIORITY=0 EVE_AMD_UPDATE_GRAPH=1" "EVE_AMD_COMM_PRIORITY=0 EVE_AMD_UPDATE_GRAPH=0" "EVE_AMD_COMM_PRIORITY=-1 EVE_AMD_UPDATE_GRAPH=0"; do
  env $var EVE_AMD_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$((PORT=PORT+1)) timeout 300 python bench.py --batch $b $Q 2>>$O/err.log | line "rccl1 B=$b gated [$var]" >> $O/sweep.txt
done
EVE_AMD_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$((PORT=PORT+1)) timeout 300 python bench.py --batch $b --graph-collectives $Q 2>>$O/err.log | line "rccl1 B=$b captured" >> $O/sweep.txt
done
python bench.py --workload c3 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>>$O/err.log | line "c3" >> $O/sweep.txt
timeout 300 python tools/aten_ops_eve.py 8 > $O/aten_ops.txt 2>>$O/err.log
cat $O/sweep.txt; head -12 $O/aten_ops.txt; tail -3 $O/err.log
