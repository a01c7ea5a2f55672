#!/bin/bash
# HBM bytes of one conv pass (separate --pmc passes):  pmc_hbm_kernel.sh <pass> IH Cin Cout K stride pad <kernel-name filter>
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmch_$c
  rocprofv3 --pmc $c -d /tmp/pmch_$c -o p --output-format csv -- python $R/tools/one_conv.py $1 $2 $3 $4 $5 $6 $7 > /dev/null 2>&1
  python $R/tools/pmcsum.py /tmp/pmch_$c "$8"
done
