cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 120 rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmc_x3/$tag -o out --output-format csv -- $R/tools/ubench/bin/gemm_x3 > /dev/null 2>$R/gpurun_out/pmc_x3/$tag.err
done
ls -R $R/gpurun_out/pmc_x3 | head -40
