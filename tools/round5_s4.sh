#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_diffusion.py tests/test_gpu_fullsize.py -q -m gpu -k "attention or forward or sampler or p_sample" 2>&1 | tail -5
timeout 600 python -m pytest tests/test_gpu_vocoder.py tests/test_gpu_e2e.py -q -m gpu -k "range_check or spurious or timeout" 2>&1 | tail -5
BB=8 bash tools/pmc_pipes.sh r05b > gpurun_out/r05b_pmc.log 2>&1
BB=8 python tools/pmc_pipes.py r05b gpurun_out/r05b_pmc_pipes.txt | head -12
