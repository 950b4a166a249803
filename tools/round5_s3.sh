#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
K="forward_T936 or p_sample_T936 or attention_block_T936"
echo "== nopk lib, old attention kernel"; DTTS_ATTN_KERNEL=w timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "$K" 2>&1 | tail -8
echo "== pk lib, old attention kernel"; DTTS_LIB_PATH=$R/detail_tts_amd/libdetail_hip_pk.so DTTS_ATTN_KERNEL=w timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "$K" 2>&1 | tail -8
echo "== pk lib, new attention kernel"; DTTS_LIB_PATH=$R/detail_tts_amd/libdetail_hip_pk.so timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "$K" 2>&1 | tail -8
echo "== nopk lib, new kernel occ 3"; DTTS_ATTN_OCC=3 timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "$K" 2>&1 | tail -8
