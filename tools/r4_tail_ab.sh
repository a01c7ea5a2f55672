#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/tail_ab
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for f in 1 0; do
  EVE_AMD_FUSE_TAIL=$f timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof$f -o b --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --batch 8 --no-cpu-baseline --no-c3 --no-c5 --no-points --no-roofline > $O/log$f.txt 2>&1
  python $R/tools/kstats.py $(find $O/prof$f -name "*kernel_stats.csv" | head -1) 6 0.004 | grep -i "linear\|gru\|eye_losses\|total\|Fill\|copy\|cat\|elementwise\|adam\|sumsq\|cast\|avgpool" > $O/k$f.txt
  rm -rf $O/prof$f
done
paste -d'\n' /dev/null; echo FUSED; cat $O/k1.txt; echo UNFUSED; cat $O/k0.txt
