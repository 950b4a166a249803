#!/bin/bash
# round 6, session 3: full GPU suite + the default bench line with infer_stream's stage A on 64 workgroups, A/B against 128
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/s3_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s3_pytest.log
tail -6 gpurun_out/s3_pytest.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/s3_bench.json 2> gpurun_out/s3_bench.err; echo "bench rc=$?"
python - <<PY
import json
d = json.loads(open("gpurun_out/s3_bench.json").read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "pipelined_equals_blocking", "unpipelined_ms_per_step", "batch1_latency_ms", "batch1_pipelined_ms_per_request", "stage_ms", "power"):
    print(k, d.get(k))
print("ragged", d.get("ragged_batch", {}).get("audio_s_per_s"), "longform", d.get("longform", {}).get("pipelined_audio_s_per_s"))
PY
for rep in 1 2 3; do
  for w in 128 64; do
    DTTS_STREAM_TOKEN_WGS=$w DTTS_BENCH_NO_EXTRA=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/s3_ab_${w}_$rep.json 2> gpurun_out/s3_ab_${w}_$rep.err
    python - $w $rep <<PY
import json, sys
w, r = sys.argv[1:]
try:
    d = json.loads(open(f"gpurun_out/s3_ab_{w}_{r}.json").read().strip().splitlines()[-1])
    p = d.get("power") or {}
    print(f"stream_token_wgs={w:4s} rep={r}: {d['ms_per_step']:7.2f} ms/step, {d['value']:.2f} audio-s/s, stage-A alone {d['stage_ms'].get('gpt_decode')} ms, diff_sample alone {d['stage_ms'].get('diff_sample')} ms, "
          f"{p.get('mean_W')} W, {p.get('mean_sclk_MHz')} MHz, {p.get('energy_J_per_step')} J/step, equal={d.get('pipelined_equals_blocking')}")
except Exception as ex:
    print(f"wgs={w} rep={r}: FAILED {ex}")
PY
  done
done 2>&1 | tee gpurun_out/s3_ab.txt
