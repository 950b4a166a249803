#!/bin/bash
# everything the round's DESIGN / profiles quote, in one GPU call:  gpurun -- 'bash tools/round3_measure.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/profile_round.sh r03 > gpurun_out/r03_profile_round.log 2>&1
BB=8 bash tools/pmc_pipes.sh r03 > gpurun_out/r03_pmc_pipes.log 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r03_default_bench.json 2> gpurun_out/r03_default_bench.err
python bench.py --steps 10 --warmup 3 --batch 1 --no-cpu-baseline > gpurun_out/r03_batch1_bench.json 2>/dev/null
DTTS_BENCH_PIPELINE=0 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r03_nopipe_bench.json 2>/dev/null
python tools/longform.py > gpurun_out/r03_longform.txt 2>&1
python tools/bench_vocoder.py > gpurun_out/r03_vocoder.txt 2>&1
DTTS_PROF_SHAPES=1 python tools/bench_layer.py > gpurun_out/r03_layer.txt 2>&1
BB=8 python tools/bench_gpt.py > gpurun_out/r03_gpt.txt 2>&1
tail -c 600 gpurun_out/r03_default_bench.json
