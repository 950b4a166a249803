#!/bin/bash
# Round profile on the GPU box: rocprofv3 kernel stats of the default bench + HBM traffic counters of one diffusion layer.
#   gpurun -- 'bash tools/profile_round.sh r01_d'      -> gpurun_out/<tag>_*   (then: python tools/profile_collect.py <tag>)
TAG=${1:-r01_x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_stats -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $OUT/${TAG}_pmc_$c -o layer --output-format csv -- python $R/tools/bench_layer.py > $OUT/${TAG}_pmc_$c.log 2>&1
done
ls $OUT | head -30
