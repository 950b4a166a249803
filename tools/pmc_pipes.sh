#!/bin/bash
# SQ pipe / wait / cache counters of the trunk kernels of one diffusion layer (tools/bench_layer.py at the merged-CFG shape B = 16):
#   gpurun -- 'bash tools/pmc_pipes.sh r02_x'  -> gpurun_out/<tag>_pmc_*  ;  then  python tools/pmc_pipes.py <tag>
TAG=${1:-r02_x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export BB=${BB:-16}
echo $BB > $OUT/${TAG}_pmc_bb.txt      # the per-launch batch, for the summary header (tools/pmc_pipes.py)
rocprofv3 -L > $OUT/${TAG}_counters_list.txt 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d $OUT/${TAG}_pmc_$i -o layer --output-format csv -- python $R/tools/bench_layer.py > $OUT/${TAG}_pmc_$i.log 2>&1
  f=$(find $OUT/${TAG}_pmc_$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && cp $f $OUT/${TAG}_pmc_$i.csv
  k=$(find $OUT/${TAG}_pmc_$i -name '*kernel_trace.csv' | head -1)
  [ -n "$k" ] && [ $i = 1 ] && cp $k $OUT/${TAG}_kernel_trace.csv
  rm -rf $OUT/${TAG}_pmc_$i
done
ls -la $OUT | grep ${TAG}
