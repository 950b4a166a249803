cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
rm -f gpurun_out/r06_measured_errors.txt
DTTS_TEST_LOG=$PWD/gpurun_out/r06_measured_errors.txt timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for rep in 1 2; do for cfg in base skip_c; do
  case $cfg in base) E="DTTS_X=0";; skip_c) E="DTTS_EXPERIMENT_SKIP_C=1";; esac
  env $E DTTS_BENCH_NO_EXTRA=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d.get('power') or {}
print('$cfg rep=$rep: %.2f ms/step, %s W, %s MHz, %s J/step' % (d['ms_per_step'], p.get('mean_W'), p.get('mean_sclk_MHz'), p.get('energy_J_per_step')))"
done; done 2>&1 | tee gpurun_out/r06_stage_c_floor.txt
