#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for pairs in 1 0; do
echo "== EVE_STEM_FWD_PAIRS=$pairs"
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA"; do
  rm -rf /tmp/pmck
  EVE_STEM_FWD_PAIRS=$pairs rocprofv3 --pmc $set -d /tmp/pmck -o p --output-format csv -- python $R/tools/one_stem.py ${1:-1920} 2 > /dev/null 2>&1
  python $R/tools/pmcsum.py /tmp/pmck "stem_fwd"
done
done
