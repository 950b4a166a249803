cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_gpt.py -x -q -k "narrow" 2>&1 | tail -2
DTTS_GPT_TOKEN_WGS=64 timeout 300 python tools/bench_gpt.py 2>&1 | tail -1
DTTS_GPT_TOKEN_WGS=64 DTTS_GPT_TOKEN_TRACE=300 timeout 300 python tools/bench_gpt.py 2>&1 >/dev/null | grep -A3 "workgroup 0" | cut -c1-400
for rep in 1 2 3; do
  DTTS_BENCH_NO_EXTRA=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d.get('power') or {}
print('rep=$rep: %.2f ms/step, %s W, %s MHz, %s J/step' % (d['ms_per_step'], p.get('mean_W'), p.get('mean_sclk_MHz'), p.get('energy_J_per_step')))"
done
