cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_gpt.py -x -q -k "narrow" 2>&1 | tail -5
for w in 128 64; do
  echo "== DTTS_GPT_TOKEN_WGS=$w"
  DTTS_GPT_TOKEN_WGS=$w timeout 300 python tools/bench_gpt.py 2>&1 | tail -1
  DTTS_GPT_TOKEN_WGS=$w timeout 300 python tools/pipeline_trace.py --requests 6 2>&1 | grep -v amdgpu.ids | tail -8
done 2>&1 | tee gpurun_out/s2c_trace.txt
