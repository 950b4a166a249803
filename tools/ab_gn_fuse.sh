#!/bin/bash
# A/B of the fused GroupNorm epilogues (DTTS_GN_FUSE) on the bench, interleaved, with 1 and 2 CFG stream chunks
mkdir -p gpurun_out
for rep in 1 2; do
  for cfg in "1 2" "0 2" "1 1" "0 1"; do
    set -- $cfg
    DTTS_GN_FUSE=$1 DTTS_CFG_STREAMS=$2 DTTS_BENCH_NO_EXTRA=1 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/ab_gn_$1_$2_$rep.json 2> gpurun_out/ab_gn_$1_$2_$rep.err
    python - "$1" "$2" "$rep" <<PY
import json, sys
f, c, r = sys.argv[1:]
try:
    d = json.loads(open(f"gpurun_out/ab_gn_{f}_{c}_{r}.json").read().strip().splitlines()[-1])
    print(f"fuse={f} cfg_streams={c} rep={r}: {d['ms_per_step']} ms/step, diff_sample {d['stage_ms'].get('diff_sample')} ms, conv frac {d['roofline']['frac']}, conv avg {d['roofline']['avg_launch_us']} us")
except Exception as e:
    print(f"fuse={f} cfg={c} rep={r}: FAILED {e}")
PY
  done
done
