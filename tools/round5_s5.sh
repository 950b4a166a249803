#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
for occ in 2 3; do
echo "== tests occ $occ"; DTTS_ATTN_OCC=$occ timeout 900 python -m pytest tests/test_gpu_diffusion.py tests/test_gpu_fullsize.py -q -m gpu -k "attention or forward or sampler or p_sample" 2>&1 | tail -4
done
DTTS_ATTN_KERNEL=w python tools/bench_layer.py 2>&1 | grep "flash_attn\|wall"
for occ in 2 3; do DTTS_ATTN_OCC=$occ python tools/bench_layer.py 2>&1 | grep "flash_attn\|wall"; done
for occ in 2 3; do BB=16 DTTS_ATTN_OCC=$occ python tools/bench_layer.py 2>&1 | grep "flash_attn\|wall"; done
for occ in 2 3; do BB=4 TT=5624 DTTS_ATTN_OCC=$occ python tools/bench_layer.py 2>&1 | grep "flash_attn\|wall"; done
DTTS_ATTN_OCC=3 BB=8 bash tools/pmc_pipes.sh r05c > gpurun_out/r05c_pmc.log 2>&1
BB=8 python tools/pmc_pipes.py r05c gpurun_out/r05c_pmc_pipes.txt | sed -n 5,12p
STEPS=8 WARMUP=3 bash tools/ab_env.sh "DTTS_ATTN_KERNEL=w" "DTTS_ATTN_OCC=3" 2 | tee gpurun_out/r05_ab_attn2.txt
