#!/bin/bash
# round-5 third GPU call: pipelined conv-GRU scan (tests + c3 / c5 timing), B = 8 dispatch sweeps, one-rank RCCL modes
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5c
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_refinenet.py -m gpu -q -x --timeout 600 -k "scan or golden or float32" 2>&1 | tail -15 > $O/pytest_scan.log
tail -6 $O/pytest_scan.log
Q="--no-cpu-baseline --no-c3 --no-c5 --no-points --no-roofline"
line() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$1', round(d['value']), 'frames/s', round(d['ms_per_step'],3), 'ms')"; }
for w in c3 c5; do python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>>$O/err.log | line "$w" >> $O/sweep.txt; done
python bench.py --batch 8 $Q 2>>$O/err.log | line "b8 default" >> $O/sweep.txt
EVE_WGRAD_HALO_MIN_M=400000 python bench.py --batch 8 $Q 2>>$O/err.log | line "b8 halo_min_m=400000" >> $O/sweep.txt
EVE_CONV_WG8_MIN_TILES=112 python bench.py --batch 8 $Q 2>>$O/err.log | line "b8 wg8_min_tiles=112" >> $O/sweep.txt
EVE_CONV_WG8_MIN_TILES=56 python bench.py --batch 8 $Q 2>>$O/err.log | line "b8 wg8_min_tiles=56" >> $O/sweep.txt
EVE_WGRAD_MIN_ROWS=768 python bench.py --batch 8 $Q 2>>$O/err.log | line "b8 wgrad_min_rows=768" >> $O/sweep.txt
EVE_WGRAD_MIN_ROWS=3072 python bench.py --batch 8 $Q 2>>$O/err.log | line "b8 wgrad_min_rows=3072" >> $O/sweep.txt
EVE_STEM_SPLIT=0 python bench.py --batch 8 $Q 2>>$O/err.log | line "b8 stem_split=0" >> $O/sweep.txt
python bench.py $Q 2>>$O/err.log | line "b32 default" >> $O/sweep.txt
EVE_WGRAD_MIN_ROWS=3072 python bench.py $Q 2>>$O/err.log | line "b32 wgrad_min_rows=3072" >> $O/sweep.txt
PORT=29617
for b in 8 32; do for mode in "--no-graph" "" "--graph-collectives"; do
  EVE_AMD_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$((PORT=PORT+1)) timeout 300 python bench.py --batch $b $mode $Q 2>>$O/err.log | line "rccl1 B=$b mode=[$mode]" >> $O/sweep.txt
done; done
cat $O/sweep.txt; tail -5 $O/err.log
