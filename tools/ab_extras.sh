#!/bin/bash
# interleaved A/B of environment settings on the bench WITH its extra measurements (un-pipelined, batch 1, ragged batch):
#   tools/ab_extras.sh "A=1" "B=2" ...   (REPS, STEPS, WARMUP from the environment)
mkdir -p gpurun_out
REPS=${REPS:-1}
for rep in $(seq 1 $REPS); do
  i=0
  for E in "$@"; do
    env $E python bench.py --steps ${STEPS:-6} --warmup ${WARMUP:-2} --no-cpu-baseline > gpurun_out/abx_${i}_$rep.json 2> gpurun_out/abx_${i}_$rep.err
    python - "$i" "$rep" "$E" <<PY
import json, sys
c, r, e = sys.argv[1:]
try:
    d = json.loads(open(f"gpurun_out/abx_{c}_{r}.json").read().strip().splitlines()[-1])
    print(f"[{e}] rep={r}: {d['ms_per_step']} ms/step; un-pipelined {d.get('unpipelined_ms_per_step')}; batch 1: latency {d.get('batch1_latency_ms')}, pipelined {d.get('batch1_pipelined_ms_per_request')} ms/request; ragged {(d.get('ragged_batch') or {}).get('ms_per_step')} ms/step; {(d.get('power') or {}).get('mean_W')} W")
except Exception as ex:
    print(f"[{e}] rep={r}: FAILED {ex}")
PY
    i=$((i+1))
  done
done
