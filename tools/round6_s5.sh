cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
for rep in 1 2; do
  i=0
  for E in "DTTS_X=0" "DTTS_GPT_TOKEN_NAP=1" "DTTS_GPT_TOKEN_NAP=4" "DTTS_GPT_TOKEN_PRIO=0" "DTTS_STAGE_A_PRIORITY=low"; do
    i=$((i+1))
    env $E DTTS_BENCH_NO_EXTRA=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/s5_${i}_$rep.json 2> gpurun_out/s5_${i}_$rep.err
    python - $i $rep "$E" <<PY
import json, sys
i, r, e = sys.argv[1:]
try:
    d = json.loads(open(f"gpurun_out/s5_{i}_{r}.json").read().strip().splitlines()[-1])
    p = d.get("power") or {}
    print(f"[{e:28s}] rep={r}: {d['ms_per_step']:7.2f} ms/step, {p.get('mean_W')} W, {p.get('mean_sclk_MHz')} MHz, {p.get('energy_J_per_step')} J/step")
except Exception as ex:
    print(f"[{e}] rep={r}: FAILED {ex}")
PY
  done
done 2>&1 | tee gpurun_out/s5_knobs.txt
