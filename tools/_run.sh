cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_IFETCH"; do
  tag=$(echo $grp | cut -c4-12)
  timeout 300 rocprofv3 --pmc $grp -d $R/gpurun_out/pmc_mt_$tag -o x --output-format csv -- python $R/tools/one_conv.py fwd 16 128 128 3 1 1 > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc $grp -d $R/gpurun_out/pmc_mt4_$tag -o x --output-format csv -- python $R/tools/one_conv.py fwd 4 512 512 3 1 1 > /dev/null 2>&1
done
cd $R
for d in gpurun_out/pmc_mt*; do echo "== $d"; python tools/pmcsum.py $d halo; done > gpurun_out/pmc_mt.txt 2>&1
rm -rf gpurun_out/pmc_mt_* gpurun_out/pmc_mt4_*
