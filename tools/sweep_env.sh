#!/bin/bash
# interleaved sweep of environment settings on the bench:  tools/sweep_env.sh "A=1" "B=2 C=3" ...   (REPS, STEPS, WARMUP from the environment)
mkdir -p gpurun_out
REPS=${REPS:-2}
for rep in $(seq 1 $REPS); do
  i=0
  for E in "$@"; do
    env $E DTTS_BENCH_NO_EXTRA=1 python bench.py --steps ${STEPS:-10} --warmup ${WARMUP:-3} --no-cpu-baseline ${ARGS:-} > gpurun_out/sweep_${i}_$rep.json 2> gpurun_out/sweep_${i}_$rep.err
    python - "$i" "$rep" "$E" <<PY
import json, sys
c, r, e = sys.argv[1:]
try:
    d = json.loads(open(f"gpurun_out/sweep_{c}_{r}.json").read().strip().splitlines()[-1])
    pw = d.get("power") or {}
    print(f"[{e}] rep={r}: {d['ms_per_step']} ms/step, {d['value']} audio-s/s, diff_sample {d['stage_ms'].get('diff_sample')} ms, conv frac {d['roofline']['frac']}, {pw.get('mean_W')} W, {pw.get('mean_sclk_MHz')} MHz")
except Exception as ex:
    print(f"[{e}] rep={r}: FAILED {ex}")
PY
    i=$((i+1))
  done
done
