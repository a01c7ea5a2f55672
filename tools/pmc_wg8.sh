#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_LDS_ADDR_CONFLICT"; do
  rm -rf /tmp/pmcw
  rocprofv3 --pmc $set -d /tmp/pmcw -o p --output-format csv -- python $R/tools/one_conv.py fwd ${1:-4} ${2:-512} ${3:-512} 3 1 1 > /dev/null 2>&1
  python $R/tools/pmcsum.py /tmp/pmcw wg8
done
