cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())"
for rep in 1 2 3; do for cfg in base c_low; do
  case $cfg in base) E="DTTS_X=0";; c_low) E="DTTS_STAGE_C_PRIORITY=low";; esac
  env $E DTTS_BENCH_NO_EXTRA=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d.get('power') or {}
print('$cfg rep=$rep: %.2f ms/step, %s W, %s MHz, %s J/step, equal=%s' % (d['ms_per_step'], p.get('mean_W'), p.get('mean_sclk_MHz'), p.get('energy_J_per_step'), d.get('pipelined_equals_blocking')))"
done; done 2>&1 | tee gpurun_out/r06_ab_stage_c_priority.txt
