set -x
for bb in 1 8; do DTTS_SAMPLER_TRACE=100 BB=$bb timeout 300 python tools/bench_gpt.py 2>&1 | grep -E 'sampler trace|G=235' | cut -c1-400; done
