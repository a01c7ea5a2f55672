#!/bin/bash
# Round-end measurement on the GPU box (run through gpurun from the repo root):  bash tools/measure_round.sh [tests]
#   gpurun_out/final/  bench.json, bench_b8.json, *_kernel_stats.csv, *_pmc_hbm_per_kernel.json (sha-stamped), pytest log
# The PMC passes are separate rocprofv3 runs with nothing but --pmc (no trace domains), as MI355X_MICROARCH.md prescribes.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
if [ "$1" = "tests" ]; then
  (cd $R && timeout 3000 python -m pytest tests -m gpu -q -x 2>&1 | tail -30 > $O/pytest_gpu.log)
fi
# kernel stats (7 steps in the EyeNet trace incl. warm-up / capture; tools/kstats.py divides by the adam_kernel count)
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_bench -o b --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-c3 --no-c5 --no-points > $O/bench_profiled.log 2>&1
cp $(find $O/prof_bench -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_c3 -o c --output-format csv -- python $R/tools/bench_eve.py --steps 5 > $O/c3_profiled.log 2>&1
cp $(find $O/prof_c3 -name "*kernel_stats.csv" | head -1) $O/eve_c3_kernel_stats.csv
rm -rf $O/prof_bench $O/prof_c3
# HBM traffic counters, one counter per pass
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c -d $O/pmc_b_$c -o p --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-c3 --no-c5 --no-points > /dev/null 2>&1
  timeout 900 rocprofv3 --pmc $c -d $O/pmc_c_$c -o p --output-format csv -- python $R/tools/bench_eve.py --steps 2 > /dev/null 2>&1
done
cd $R
python tools/pmc_hbm_summary.py $O/pmc_b_FETCH_SIZE $O/pmc_b_WRITE_SIZE bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-c3 --no-c5 --no-points > $O/pmc_hbm_per_kernel.json
python tools/pmc_hbm_summary.py $O/pmc_c_FETCH_SIZE $O/pmc_c_WRITE_SIZE tools/bench_eve.py --steps 2 > $O/c3_pmc_hbm_per_kernel.json
rm -rf $O/pmc_b_* $O/pmc_c_*
# the bench lines proper (un-profiled); the fresh PMC summaries are put where bench.py looks for them
RN=${ROUND:-r06}
cp $O/pmc_hbm_per_kernel.json profiles/${RN}_pmc_hbm_per_kernel.json
cp $O/c3_pmc_hbm_per_kernel.json profiles/${RN}_c3_pmc_hbm_per_kernel.json
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --batch 8 --no-cpu-baseline --no-c3 --no-c5 --no-points > $O/bench_b8.json 2>> $O/bench.err
python bench.py --dtype fp16 --no-cpu-baseline --no-c3 --no-c5 --no-points > $O/bench_fp16.json 2>> $O/bench.err
# configs[4] as its own line + kernel stats (T = 120, 256 x 256, fp16, both networks trained, B = 8)
python bench.py --workload c5 --steps 3 --warmup 2 --no-cpu-baseline > $O/bench_c5.json 2>> $O/bench.err
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_c5 -o c --output-format csv -- python $R/bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/c5_profiled.log 2>&1)
cp $(find $O/prof_c5 -name "*kernel_stats.csv" | head -1) $O/c5_kernel_stats.csv; rm -rf $O/prof_c5
# one-rank RCCL group: eager launches vs hipGraph replay + eager collectives vs collectives captured, B = 8 and B = 32
PORT=29517
for b in 8 32; do for mode in "--no-graph" "" "--graph-collectives"; do
  echo "B=$b mode=[$mode] $(EVE_AMD_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$((PORT=PORT+1)) python bench.py --batch $b $mode --no-cpu-baseline --no-c3 --no-c5 --no-points --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(round(d['value']), 'frames/s', round(d['ms_per_step'],3), 'ms', 'hip_graph', d['hip_graph'], 'collectives:', d['collectives'])")" >> $O/rccl_one_rank_modes.txt
done; done
# copies for profiles/ (gpurun_out/ is scratch; the caller commits profiles/ after the call)
mkdir -p $O/profiles
cp $O/bench.json $O/profiles/${RN}_bench.json
cp $O/bench_b8.json $O/profiles/${RN}_bench_b8.json
cp $O/bench_fp16.json $O/profiles/${RN}_bench_fp16.json
cp $O/bench_c5.json $O/profiles/${RN}_bench_c5.json
cp $O/c5_kernel_stats.csv $O/profiles/${RN}_c5_kernel_stats.csv
cp $O/rccl_one_rank_modes.txt $O/profiles/${RN}_rccl_one_rank_modes.txt
cp $O/bench_kernel_stats.csv $O/profiles/${RN}_bench_kernel_stats.csv
cp $O/eve_c3_kernel_stats.csv $O/profiles/${RN}_eve_c3_kernel_stats.csv
cp $O/pmc_hbm_per_kernel.json $O/profiles/${RN}_pmc_hbm_per_kernel.json
cp $O/c3_pmc_hbm_per_kernel.json $O/profiles/${RN}_c3_pmc_hbm_per_kernel.json
[ -f $O/pytest_gpu.log ] && cp $O/pytest_gpu.log $O/profiles/${RN}_pytest_gpu.log
# what the 16-bit format costs on a trained network, the CPU baseline at several thread counts, the ATen launch census of configs[2]
timeout 900 python tools/train_sanity.py 400 bf16 > $O/profiles/${RN}_train_sanity.log 2>&1
timeout 600 python tools/cpu_baseline_threads.py 50 > $O/profiles/${RN}_cpu_baseline_threads.log 2>&1
timeout 300 python tools/aten_ops_eve.py 8 > $O/profiles/${RN}_aten_ops_c3.txt 2>/dev/null
