#!/bin/bash
# round 5, GPU session 2: the block-skewed attention kernel (attention_x3b.hip) - parity, then timing against the round-4 kernel
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_diffusion.py tests/test_gpu_fullsize.py -q -m gpu -k "attention or forward or sampler or p_sample" > gpurun_out/r05_attn_tests.log 2>&1
tail -n 15 gpurun_out/r05_attn_tests.log
DTTS_ATTN_KERNEL=w python tools/bench_layer.py > gpurun_out/r05_layer_w.txt 2>&1
python tools/bench_layer.py > gpurun_out/r05_layer_b2.txt 2>&1
grep -h "flash_attn\|wall" gpurun_out/r05_layer_w.txt gpurun_out/r05_layer_b2.txt
BB=16 DTTS_ATTN_KERNEL=w python tools/bench_layer.py 2>&1 | grep "flash_attn\|wall"
BB=16 python tools/bench_layer.py 2>&1 | grep "flash_attn\|wall"
BB=4 TT=5624 DTTS_ATTN_KERNEL=w python tools/bench_layer.py 2>&1 | grep "flash_attn\|wall"
BB=4 TT=5624 python tools/bench_layer.py 2>&1 | grep "flash_attn\|wall"
STEPS=8 WARMUP=3 bash tools/ab_env.sh "DTTS_ATTN_KERNEL=w" "DTTS_ATTN_KERNEL=b" 2 | tee gpurun_out/r05_ab_attn.txt
