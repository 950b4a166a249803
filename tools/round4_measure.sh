#!/bin/bash
# everything the round's DESIGN / profiles quote, in one GPU call:  gpurun -- 'bash tools/round4_measure.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/profile_round.sh r04 > gpurun_out/r04_profile_round.log 2>&1
BB=8 bash tools/pmc_pipes.sh r04 > gpurun_out/r04_pmc_pipes.log 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_default_bench.json 2> gpurun_out/r04_default_bench.err
python bench.py --steps 10 --warmup 3 --batch 1 --no-cpu-baseline > gpurun_out/r04_batch1_bench.json 2>/dev/null
DTTS_BENCH_PIPELINE=0 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r04_nopipe_bench.json 2>/dev/null
python tools/longform.py > gpurun_out/r04_longform.txt 2>&1
python tools/bench_vocoder.py > gpurun_out/r04_vocoder.txt 2>&1
DTTS_PROF_SHAPES=1 python tools/bench_layer.py > gpurun_out/r04_layer.txt 2>&1
python tools/bench_forward.py > gpurun_out/r04_forward.txt 2>&1
DTTS_GN_FUSE=1 python tools/bench_forward.py > gpurun_out/r04_forward_gn_fuse.txt 2>&1
DTTS_CONV_ABLATE=512 python tools/bench_forward.py > gpurun_out/r04_forward_generic_kloop.txt 2>&1
BB=8 python tools/bench_gpt.py > gpurun_out/r04_gpt.txt 2>&1
python tools/pipeline_trace.py --requests 8 > gpurun_out/r04_pipeline_trace.txt 2>&1
tail -c 600 gpurun_out/r04_default_bench.json
