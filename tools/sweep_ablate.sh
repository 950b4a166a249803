#!/bin/bash
# Where do the ~35 ms go that stage A costs stage B per request?  The token kernel with parts of its work removed (results are garbage,
# the codes' count is fixed by suppress_eos, so the pipeline's timing stays valid), interleaved on the bench:
#   base        the shipped kernel
#   no_weights  DTTS_GPT_TOKEN_ABLATE=1: no weight loads - the 128 workgroups' hops and CU hold as they are, 308 MB per token less traffic
#   no_wait     DTTS_GPT_TOKEN_ABLATE=2: no waiting at the exchanges - the weight bytes as they are, the hold shrinks to the compute
#   neither     DTTS_GPT_TOKEN_ABLATE=3
#   exit        DTTS_GPT_TOKEN_ABLATE=4: the launch alone (128 workgroups x 160 KB LDS acquired and released at once)
#   chain       DTTS_GPT_TOKEN_KERNEL=0: the 53-launch chain
REPS="${1:-2}"
mkdir -p gpurun_out
for rep in $(seq 1 $REPS); do
  for cfg in base no_weights no_wait neither exit chain; do
    case $cfg in
      base) E="DTTS_X=0";; no_weights) E="DTTS_GPT_TOKEN_ABLATE=1";; no_wait) E="DTTS_GPT_TOKEN_ABLATE=2";; neither) E="DTTS_GPT_TOKEN_ABLATE=3";;
      exit) E="DTTS_GPT_TOKEN_ABLATE=4";; chain) E="DTTS_GPT_TOKEN_KERNEL=0";;
    esac
    env $E DTTS_BENCH_NO_EXTRA=1 python bench.py --steps ${STEPS:-8} --warmup ${WARMUP:-3} --no-cpu-baseline > gpurun_out/abl_${cfg}_$rep.json 2> gpurun_out/abl_${cfg}_$rep.err
    python - "$cfg" "$rep" <<PY
import json, sys
c, r = sys.argv[1:]
try:
    d = json.loads(open(f"gpurun_out/abl_{c}_{r}.json").read().strip().splitlines()[-1])
    p = d.get("power") or {}
    print(f"{c:11s} rep={r}: {d['ms_per_step']:7.2f} ms/step, stage-A alone {d['stage_ms'].get('gpt_decode')} ms, diff_sample alone {d['stage_ms'].get('diff_sample')} ms, "
          f"{p.get('mean_W')} W, {p.get('mean_sclk_MHz')} MHz, {p.get('energy_J_per_step')} J/step")
except Exception as ex:
    print(f"{c} rep={r}: FAILED {ex}")
PY
  done
done
