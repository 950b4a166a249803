set -x
timeout 900 python -m pytest tests/test_gpu_gpt.py -x -q -m gpu 2>&1 | tail -3
for h in 1 2; do DTTS_GPT_TOKEN_HALVES=$h BB=8 timeout 300 python tools/bench_gpt.py 2>&1 | grep 'G='; done
DTTS_GPT_TOKEN_HALVES=2 DTTS_GPT_TOKEN_TRACE=60 BB=8 timeout 300 python tools/bench_gpt.py 2>&1 | grep -A3 'workgroup 0' | cut -c1-330
STEPS=8 WARMUP=3 bash tools/ab_env.sh "X=0" "DTTS_GPT_TOKEN_HALVES=2" 3
