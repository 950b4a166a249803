cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_gpt.py -x -q -k "narrow" 2>&1 | tail -2
for w in 64 32; do
  DTTS_GPT_TOKEN_WGS=$w timeout 300 python tools/bench_gpt.py 2>&1 | tail -1
done
DTTS_GPT_TOKEN_WGS=64 DTTS_GPT_TOKEN_TRACE=300 timeout 300 python tools/bench_gpt.py 2>&1 >/dev/null | grep -A3 "workgroup 0" | cut -c1-400
DTTS_GPT_TOKEN_ABLATE=8 DTTS_GPT_TOKEN_WGS=64 DTTS_GPT_TOKEN_TRACE=300 timeout 300 python tools/bench_gpt.py 2>&1 >/dev/null | grep -A3 "workgroup 0" | cut -c1-400
