cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/s12_bench.json 2> gpurun_out/s12_bench.err; echo "bench rc=$?"
python - <<PY
import json
d = json.loads(open("gpurun_out/s12_bench.json").read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "pipelined_equals_blocking", "unpipelined_ms_per_step", "batch1_latency_ms", "stage_ms"):
    print(k, d.get(k))
print("power", {k: v for k, v in d["power"].items() if k in ("mean_W", "mean_sclk_MHz", "energy_J_per_step")})
print("roofline", {k: d["roofline"].get(k) for k in ("frac", "traffic", "traffic_over_algorithmic", "busy_share_of_timed_region")})
PY
