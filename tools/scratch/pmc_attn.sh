cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_attn
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmc_attn/s$i -o out --output-format csv -- python $R/tools/bench_layer.py > /dev/null 2>$R/gpurun_out/pmc_attn/s$i.err
done
