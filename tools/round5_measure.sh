#!/bin/bash
# everything the round's DESIGN / profiles quote, in one GPU call:  gpurun -- 'bash tools/round5_measure.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/profile_round.sh r05 > gpurun_out/r05_profile_round.log 2>&1
BB=8 bash tools/pmc_pipes.sh r05 > gpurun_out/r05_pmc_pipes.log 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r05_default_bench.json 2> gpurun_out/r05_default_bench.err
python bench.py --steps 10 --warmup 3 --batch 1 --no-cpu-baseline > gpurun_out/r05_batch1_bench.json 2>/dev/null
DTTS_BENCH_PIPELINE=0 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r05_nopipe_bench.json 2>/dev/null
DTTS_ATTN_KERNEL=w DTTS_BENCH_NO_EXTRA=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r05_bench_old_attention.json 2>/dev/null
python tools/longform.py > gpurun_out/r05_longform.txt 2>&1
python tools/bench_vocoder.py > gpurun_out/r05_vocoder.txt 2>&1
DTTS_PROF_SHAPES=1 python tools/bench_layer.py > gpurun_out/r05_layer.txt 2>&1
python tools/bench_forward.py > gpurun_out/r05_forward.txt 2>&1
BB=1 python tools/bench_forward.py > gpurun_out/r05_forward_batch1.txt 2>&1
BB=1 DTTS_GN_FUSE=1 python tools/bench_forward.py > gpurun_out/r05_forward_batch1_gn_fuse.txt 2>&1
BB=8 python tools/bench_gpt.py > gpurun_out/r05_gpt.txt 2>&1
python tools/pipeline_trace.py --requests 8 > gpurun_out/r05_pipeline_trace.txt 2>&1
tail -c 700 gpurun_out/r05_default_bench.json
