#!/bin/bash
# Round profile on the GPU box: rocprofv3 kernel stats of the DRIVER's bench command (summarised on the box: the raw trace is
# > 64 MiB), HBM traffic counters of one diffusion layer at the bench's per-launch shape (B = 8 per CFG stream).
#   gpurun -- 'bash tools/profile_round.sh r02'      -> gpurun_out/<tag>_*   (then: python tools/profile_collect.py <tag>)
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 1500 rocprofv3 --kernel-trace --stats -d /tmp/${TAG}_stats -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
DB=$(find /tmp/${TAG}_stats -name '*.db' | head -1)
python $R/tools/rocprof_summary.py $DB $OUT/${TAG}_bench_kernel_stats.txt "${TAG}: rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline (the driver's command; 27 passes: 5 warm-up, 1 stage-timing, 20 timed, + 1 vocoder-only pass under the event profiler; batch 8; the timed passes are software-pipelined (stage A of batch i + 1 and stage C of batch i run under stage B of batch i + 1 / i on their own streams) and the cond / uncond halves of every diffusion forward run on two HIP streams, so launch durations include time shared with co-running kernels)"
rm -rf /tmp/${TAG}_stats
export BB=8
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/${TAG}_pmc_$c -o layer --output-format csv -- python $R/tools/bench_layer.py > $OUT/${TAG}_pmc_$c.log 2>&1
  mkdir -p $OUT/${TAG}_pmc_$c
  cp $(find /tmp/${TAG}_pmc_$c -name '*counter_collection.csv' | head -1) $OUT/${TAG}_pmc_$c/layer_counter_collection.csv
  rm -rf /tmp/${TAG}_pmc_$c
done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/${TAG}_pmcvoc_$c -o voc --output-format csv -- python $R/tools/bench_vocoder.py > $OUT/${TAG}_pmcvoc_$c.log 2>&1
  mkdir -p $OUT/${TAG}_pmcvoc_$c
  python - <<PY
import csv, glob
f = glob.glob("/tmp/${TAG}_pmcvoc_$c/**/*counter_collection.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "resblock1x3_fused" in r["Kernel_Name"]]
w = csv.DictWriter(open("$OUT/${TAG}_pmcvoc_$c/voc_counter_collection.csv", "w"), fieldnames=list(rows[0].keys()))
w.writeheader(); w.writerows(rows)
PY
  rm -rf /tmp/${TAG}_pmcvoc_$c
done
du -sh $OUT
