#!/bin/bash
# round 5, GPU session 1: co-residency gates (tests/test_gpu_hazard.py) with the default library and with the whole library built without
# packed fp32 instructions (libdetail_hip_nopk.so: -Xclang -target-feature -Xclang -packed-fp32-ops), and what that build costs.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
NOPK=$R/detail_tts_amd/libdetail_hip_nopk.so
timeout 900 python -m pytest tests/test_gpu_hazard.py -q -m gpu > gpurun_out/r05_hazard_default.log 2>&1
DTTS_LIB_PATH=$NOPK timeout 900 python -m pytest tests/test_gpu_hazard.py -q -m gpu > gpurun_out/r05_hazard_nopk.log 2>&1
tail -3 gpurun_out/r05_hazard_default.log gpurun_out/r05_hazard_nopk.log
STEPS=8 WARMUP=3 bash tools/ab_env.sh "DTTS_AB=default" "DTTS_LIB_PATH=$NOPK" 2 | tee gpurun_out/r05_ab_nopk.txt
python tools/bench_forward.py > gpurun_out/r05_forward_default.txt 2>&1
DTTS_LIB_PATH=$NOPK python tools/bench_forward.py > gpurun_out/r05_forward_nopk.txt 2>&1
BB=8 python tools/bench_gpt.py > gpurun_out/r05_gpt_default.txt 2>&1
DTTS_LIB_PATH=$NOPK BB=8 python tools/bench_gpt.py > gpurun_out/r05_gpt_nopk.txt 2>&1
python tools/bench_vocoder.py > gpurun_out/r05_vocoder_default.txt 2>&1
DTTS_LIB_PATH=$NOPK python tools/bench_vocoder.py > gpurun_out/r05_vocoder_nopk.txt 2>&1
tail -4 gpurun_out/r05_forward_default.txt gpurun_out/r05_forward_nopk.txt
