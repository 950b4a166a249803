cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
for rep in 1 2 3; do for cfg in base b_high; do
  case $cfg in base) E="DTTS_X=0";; b_high) E="DTTS_STAGE_B_PRIORITY=high DTTS_B_PRIORITY=high";; esac
  env $E DTTS_BENCH_NO_EXTRA=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>gpurun_out/s10_$cfg.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d.get('power') or {}
print('$cfg rep=$rep: %.2f ms/step, %s W, %s MHz, %s J/step, equal=%s' % (d['ms_per_step'], p.get('mean_W'), p.get('mean_sclk_MHz'), p.get('energy_J_per_step'), d.get('pipelined_equals_blocking')))"
done; done 2>&1 | tee gpurun_out/r06_ab_stage_b_priority.txt
tail -3 gpurun_out/s10_b_high.err
