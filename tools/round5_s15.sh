#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "fused_groupnorm or forward_T936" 2>&1 | tail -6
for f in 0 1; do echo "B=1 fuse=$f"; BB=1 DTTS_GN_FUSE=$f python tools/bench_forward.py 2>&1 | tail -10; done
for f in 0 1; do echo "B=2 fuse=$f"; BB=2 DTTS_GN_FUSE=$f python tools/bench_forward.py 2>&1 | tail -2; done
for f in 0 1; do DTTS_GN_FUSE=$f DTTS_BENCH_NO_EXTRA=1 DTTS_BENCH_PIPELINE=0 python bench.py --steps 5 --warmup 2 --batch 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch1 fuse=$f', d['ms_per_step'], d['stage_ms'])"; done
