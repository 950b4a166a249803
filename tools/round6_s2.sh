#!/bin/bash
# round 6, session 2: the 64 / 32-workgroup token kernels (gpt_token_n.hip) - parity, decode time alone, phase stamps, bench A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_gpt.py -x -q -k "narrow or token_kernel_equals or one_and_four" > gpurun_out/s2_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s2_pytest.log
tail -15 gpurun_out/s2_pytest.log
for w in 128 64 32; do
  echo "== DTTS_GPT_TOKEN_WGS=$w"
  DTTS_GPT_TOKEN_WGS=$w timeout 300 python tools/bench_gpt.py 2>&1 | tail -2
done > gpurun_out/s2_decode.txt 2>&1
cat gpurun_out/s2_decode.txt
for w in 64 32; do
  DTTS_GPT_TOKEN_WGS=$w DTTS_GPT_TOKEN_TRACE=300 timeout 300 python tools/bench_gpt.py > /dev/null 2> gpurun_out/s2_trace_$w.txt
  grep -A3 "workgroup 0" gpurun_out/s2_trace_$w.txt | cut -c1-400
done
for rep in 1 2; do
  for w in 128 64 32; do
    DTTS_GPT_TOKEN_WGS=$w DTTS_BENCH_NO_EXTRA=1 timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/s2_ab_${w}_$rep.json 2> gpurun_out/s2_ab_${w}_$rep.err
    python - $w $rep <<PY
import json, sys
w, r = sys.argv[1:]
try:
    d = json.loads(open(f"gpurun_out/s2_ab_{w}_{r}.json").read().strip().splitlines()[-1])
    p = d.get("power") or {}
    print(f"wgs={w:4s} rep={r}: {d['ms_per_step']:7.2f} ms/step, stage-A alone {d['stage_ms'].get('gpt_decode')} ms, diff_sample alone {d['stage_ms'].get('diff_sample')} ms, "
          f"{p.get('mean_W')} W, {p.get('mean_sclk_MHz')} MHz, {p.get('energy_J_per_step')} J/step, equal={d.get('pipelined_equals_blocking')}, unpipelined {d.get('unpipelined_ms_per_step')}")
except Exception as ex:
    print(f"wgs={w} rep={r}: FAILED {ex}")
PY
  done
done 2>&1 | tee gpurun_out/s2_ab.txt
