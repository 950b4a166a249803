#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
echo "== tests occ 3"; timeout 900 python -m pytest tests/test_gpu_diffusion.py tests/test_gpu_fullsize.py -q -m gpu -k "attention or forward or p_sample" 2>&1 | tail -3
echo "old kernel:"; DTTS_ATTN_KERNEL=w python tools/bench_layer.py 2>&1 | grep "flash_attn\|wall"
for occ in 3 2; do echo "new occ $occ:"; DTTS_ATTN_OCC=$occ python tools/bench_layer.py 2>&1 | grep "flash_attn\|wall"; done
for a in 1 2 3 4 8 16 24 28 32 35 39; do echo "ablate $a:"; DTTS_ATTN_ABLATE=$a python tools/bench_layer.py 2>&1 | grep "flash_attn"; done
