#!/bin/bash
O=gpurun_out/rccl; mkdir -p $O; rm -f $O/modes.txt
PORT=29517
for b in 8 32; do for mode in "--no-graph" "" "--graph-collectives"; do
  EVE_AMD_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$((PORT=PORT+1)) python bench.py --batch $b $mode --no-cpu-baseline --no-c3 --no-c5 --no-points --no-roofline > $O/out.json 2> $O/err_${b}_$(echo $mode | tr -d " -").log
  echo "B=$b mode=[$mode] $(python -c "import sys,json; d=json.loads([l for l in open('$O/out.json') if l.startswith('{')][0]); print(round(d['value']), 'frames/s', round(d['ms_per_step'],3), 'ms', 'hip_graph', d['hip_graph'], 'collectives:', d['collectives'])" 2>&1 | tail -1)" >> $O/modes.txt
done; done
cat $O/modes.txt; tail -5 $O/err.log | cut -c1-300
