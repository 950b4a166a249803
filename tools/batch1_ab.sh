#!/bin/bash
# batch-1 (configs[1]) A/B of environment settings: tools/batch1_ab.sh "NAME=VALUE ..." "NAME=VALUE ..." ...   (one bench run per setting, REPS rounds)
# prints the blocking stage times (stage_ms: the latency components) and the pipelined period per request
mkdir -p gpurun_out
REPS="${REPS:-1}"
for rep in $(seq 1 $REPS); do
  i=0
  for E in "$@"; do
    i=$((i + 1))
    env $E DTTS_BENCH_NO_EXTRA=1 python bench.py --batch 1 --steps ${STEPS:-8} --warmup ${WARMUP:-3} --no-cpu-baseline > gpurun_out/b1_${i}_$rep.json 2> gpurun_out/b1_${i}_$rep.err
    python - "$i" "$rep" "$E" <<PY
import json, sys
c, r, e = sys.argv[1:]
try:
    d = json.loads(open(f"gpurun_out/b1_{c}_{r}.json").read().strip().splitlines()[-1])
    st = d["stage_ms"]
    print(f"[{e}] rep={r}: latency (sum of blocking stages) {sum(st.values()):.1f} ms = {st}; pipelined {d['ms_per_step']} ms/request; power {d.get('power', {}).get('mean_W')} W")
except Exception as ex:
    print(f"[{e}] rep={r}: FAILED {ex}")
PY
  done
done
