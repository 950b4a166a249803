set -x
timeout 900 python -m pytest tests/test_gpu_gpt.py -x -q -m gpu 2>&1 | tail -3
for mr in 1 4 8; do DTTS_GPT_TOKEN_MIN_ROWS=$mr BB=1 timeout 300 python tools/bench_gpt.py 2>&1 | grep 'G=235'; done
DTTS_GPT_TOKEN_TRACE=60 BB=1 timeout 300 python tools/bench_gpt.py 2>&1 | grep -A12 'workgroup 0' | cut -c1-330
REPS=2 tools/batch1_ab.sh "X=0" "DTTS_GPT_TOKEN_MIN_ROWS=4"
