set -x
timeout 900 python -m pytest tests/test_gpu_gpt.py -x -q -m gpu -k "token or sixteen" 2>&1 | tail -5
for mr in 4 8; do for bb in 1 4; do DTTS_GPT_TOKEN_MIN_ROWS=$mr BB=$bb timeout 300 python tools/bench_gpt.py 2>&1 | grep 'G='; done; done
DTTS_GPT_TOKEN_TRACE=60 BB=1 timeout 300 python tools/bench_gpt.py 2>&1 | grep -A14 'workgroup 0' | head -16
REPS=2 tools/batch1_ab.sh "X=0" "DTTS_CFG_STREAMS=2" "DTTS_CFG_STREAMS=2 DTTS_CONV_KSPLIT=8 DTTS_CONV_KSPLIT_K1=4" "DTTS_CONV_KSPLIT_K1=4" "DTTS_GPT_TOKEN_MIN_ROWS=8"
