#!/bin/bash
# interleaved A/B of environment settings on the bench: tools/ab_env.sh "NAME=VALUE ..." "NAME=VALUE ..." [reps]
A="$1"; B="$2"; REPS="${3:-2}"
mkdir -p gpurun_out
for rep in $(seq 1 $REPS); do
  for cfg in A B; do
    if [ $cfg = A ]; then E="$A"; else E="$B"; fi
    env $E DTTS_BENCH_NO_EXTRA=1 python bench.py --steps ${STEPS:-8} --warmup ${WARMUP:-3} --no-cpu-baseline > gpurun_out/ab_env_${cfg}_$rep.json 2> gpurun_out/ab_env_${cfg}_$rep.err
    python - "$cfg" "$rep" "$E" <<PY
import json, sys
c, r, e = sys.argv[1:]
try:
    d = json.loads(open(f"gpurun_out/ab_env_{c}_{r}.json").read().strip().splitlines()[-1])
    print(f"{c} [{e}] rep={r}: {d['ms_per_step']} ms/step, {d['value']} audio-s/s, diff_sample {d['stage_ms'].get('diff_sample')} ms, conv frac {d['roofline']['frac']}, conv avg {d['roofline']['avg_launch_us']} us")
except Exception as ex:
    print(f"{c} rep={r}: FAILED {ex}")
PY
  done
done
