#!/bin/bash
# GPT stage (tools/bench_gpt.py): timing sweep over the launch mode / key splits, then a rocprofv3 kernel trace of the default and
# of the eager mode with per-kernel durations and gaps.      gpurun -- 'bash tools/prof_gpt.sh tag'
TAG=${1:-gpt}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in "DTTS_GPT_GRAPH=1" "DTTS_GPT_GRAPH=0" "DTTS_GPT_GRAPH_CHUNK=4" "DTTS_GPT_GRAPH_CHUNK=64" "DTTS_GPT_KSPLIT=1" "DTTS_GPT_KSPLIT=4" $EXTRA_CFGS; do
  echo "== ${cfg:-default}"; env $cfg python $R/tools/bench_gpt.py 2>&1 | grep "G=235"
done
for mode in graph eager; do
  if [ $mode = eager ]; then export DTTS_GPT_GRAPH=0; else export DTTS_GPT_GRAPH=1; fi
  timeout 600 rocprofv3 --kernel-trace -d $OUT/${TAG}_trace_$mode -o gpt -- python $R/tools/bench_gpt.py > $OUT/${TAG}_trace_$mode.log 2>&1
  DB=$(find $OUT/${TAG}_trace_$mode -name '*.db' | head -1)
  echo "== trace $mode"; python $R/tools/kernel_timeline.py $DB "" 12402 | tee $OUT/${TAG}_timeline_$mode.txt
  rm -rf $OUT/${TAG}_trace_$mode
done
