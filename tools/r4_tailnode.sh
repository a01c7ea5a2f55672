#!/bin/bash
# the one-node tail: parity tests, then the step at B=32 and B=8 with and without it
mkdir -p gpurun_out
python -m pytest tests/test_gpu_eyenet.py -x -q -k "one_node or trainer_takes or tail_chains" > gpurun_out/tailnode_tests.log 2>&1
tail -5 gpurun_out/tailnode_tests.log
for node in 1 0; do
  for b in 32 8; do
    echo "node=$node B=$b" 
    EVE_AMD_TAIL_LOSS_NODE=$node python bench.py --steps 30 --warmup 10 --batch $b --no-c3 --no-c5 --no-points --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done
done 2>&1 | tee gpurun_out/tailnode_ab.txt
