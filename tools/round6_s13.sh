cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
for rep in 1 2; do for cfg in 2 1 4; do
  DTTS_CFG_STREAMS=$cfg DTTS_BENCH_NO_EXTRA=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d.get('power') or {}
print('cfg_streams=$cfg rep=$rep: %.2f ms/step, %s W, %s MHz, %s J/step' % (d['ms_per_step'], p.get('mean_W'), p.get('mean_sclk_MHz'), p.get('energy_J_per_step')))"
done; done 2>&1 | tee gpurun_out/r06_ab_cfg_streams.txt
