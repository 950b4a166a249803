cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
for rep in 1 2; do
  for cfg in 128 64 64x; do
    case $cfg in 128) E="DTTS_GPT_TOKEN_WGS=128";; 64) E="DTTS_GPT_TOKEN_WGS=64";; 64x) E="DTTS_GPT_TOKEN_WGS=64 DTTS_GPT_TOKEN_EXCLUSIVE_CU=1";; esac
    env $E DTTS_BENCH_NO_EXTRA=1 timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/s2f_ab_${cfg}_$rep.json 2> gpurun_out/s2f_ab_${cfg}_$rep.err
    python - $cfg $rep <<PY
import json, sys
w, r = sys.argv[1:]
try:
    d = json.loads(open(f"gpurun_out/s2f_ab_{w}_{r}.json").read().strip().splitlines()[-1])
    p = d.get("power") or {}
    print(f"cfg={w:4s} rep={r}: {d['ms_per_step']:7.2f} ms/step, stage-A alone {d['stage_ms'].get('gpt_decode')} ms, diff_sample alone {d['stage_ms'].get('diff_sample')} ms, "
          f"{p.get('mean_W')} W, {p.get('mean_sclk_MHz')} MHz, {p.get('energy_J_per_step')} J/step")
except Exception as ex:
    print(f"cfg={w} rep={r}: FAILED {ex}")
PY
  done
done 2>&1 | tee gpurun_out/s2f_ab.txt
