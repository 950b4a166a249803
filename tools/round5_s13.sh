#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
DTTS_BENCH_NO_EXTRA=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r05_bench_power.json 2>gpurun_out/r05_bench_power.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05_bench_power.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["power"], d["pipelined_equals_blocking"])
PY
DTTS_ATTN_KERNEL=w python tools/longform.py > gpurun_out/r05_longform_w.txt 2>&1
python tools/longform.py > gpurun_out/r05_longform_b.txt 2>&1
tail -6 gpurun_out/r05_longform_w.txt gpurun_out/r05_longform_b.txt
DTTS_BENCH_NO_EXTRA=1 DTTS_BENCH_PIPELINE=0 python bench.py --steps 5 --warmup 2 --batch 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch1', d['ms_per_step'], d['stage_ms'], d['power'])"
