#!/bin/bash
# round-5 measurement set: profiles/r05_* (kernel stats, PMC traffic, bench lines, one-rank RCCL modes), trained-network
# bf16 cost, CPU-baseline thread sweep
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
ROUND=r05 bash tools/measure_round.sh
O=$R/gpurun_out/final
timeout 900 python tools/train_sanity.py 400 bf16 > $O/profiles/r05_train_sanity.log 2>&1
timeout 600 python tools/cpu_baseline_threads.py 50 > $O/profiles/r05_cpu_baseline_threads.log 2>&1
timeout 300 python tools/aten_ops_eve.py 8 > $O/profiles/r05_aten_ops_c3.txt 2>/dev/null
cut -c1-600 $O/bench.json; cat $O/rccl_one_rank_modes.txt; tail -14 $O/profiles/r05_train_sanity.log | head -6; cat $O/profiles/r05_cpu_baseline_threads.log; tail -3 $O/bench.err
