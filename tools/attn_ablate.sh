#!/bin/bash
# the attention kernel's ablation table (profiles/r05_attn_ablate_pipelined.txt):  gpurun -- 'bash tools/attn_ablate.sh'
# DTTS_ATTN_ABLATE bits: 1 no LDS-DMA in the loop, 2 no barrier / DMA wait, 4 no softmax vector work, 8 no PV MFMAs, 16 no QK^T MFMAs, 32 no LDS fragment reads
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
echo "round 2-4 kernel:"; DTTS_ATTN_KERNEL=w python tools/bench_layer.py 2>&1 | grep "flash_attn\|wall"
for occ in 2 3; do echo "attention_x3b, $occ workgroups per CU:"; DTTS_ATTN_OCC=$occ python tools/bench_layer.py 2>&1 | grep "flash_attn\|wall"; done
for a in 1 2 3 4 8 16 24 28 32 35 39; do echo "ablate $a:"; DTTS_ATTN_ABLATE=$a python tools/bench_layer.py 2>&1 | grep "flash_attn"; done
