cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
for w in 64; do
  DTTS_GPT_TOKEN_ABLATE=8 DTTS_GPT_TOKEN_WGS=$w DTTS_GPT_TOKEN_TRACE=300 timeout 300 python tools/bench_gpt.py > /dev/null 2> gpurun_out/s2_atrace_$w.txt
  grep -A4 "workgroup 0" gpurun_out/s2_atrace_$w.txt | cut -c1-400
done
