# usage: bash tools/pmc_one_conv.sh "<W8 values>"   -- PMC passes (SQ only, kernel-trace) of tools/one_conv.py per forced kernel choice
cd /root/repo
export TMPDIR=/tmp
for cfg in ${1:-0 -2}; do
 for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | cut -c1-12 | tr ' ' '_')
  W8=$cfg REPS=10 rocprofv3 --pmc $pass --kernel-trace -d gpurun_out/pmc_${cfg}_$tag --output-format csv -- python3 tools/one_conv.py > gpurun_out/pmc_${cfg}_$tag.log 2>&1
 done
done
