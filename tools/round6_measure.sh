#!/bin/bash
# everything the round's DESIGN / profiles quote, in one GPU call:  gpurun -- 'bash tools/round6_measure.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
bash tools/profile_round.sh r06 > gpurun_out/r06_profile_round.log 2>&1
BB=8 bash tools/pmc_pipes.sh r06 > gpurun_out/r06_pmc_pipes.log 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r06_default_bench.json 2> gpurun_out/r06_default_bench.err
python bench.py --steps 10 --warmup 3 --batch 1 --no-cpu-baseline > gpurun_out/r06_batch1_bench.json 2>/dev/null
DTTS_BENCH_PIPELINE=0 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r06_nopipe_bench.json 2>/dev/null
python tools/longform.py > gpurun_out/r06_longform.txt 2>&1
python tools/bench_vocoder.py > gpurun_out/r06_vocoder.txt 2>&1
DTTS_PROF_SHAPES=1 python tools/bench_layer.py > gpurun_out/r06_layer.txt 2>&1
python tools/bench_forward.py > gpurun_out/r06_forward.txt 2>&1
BB=1 python tools/bench_forward.py > gpurun_out/r06_forward_batch1.txt 2>&1
for w in 128 64 32; do echo "== DTTS_GPT_TOKEN_WGS=$w"; DTTS_GPT_TOKEN_WGS=$w BB=8 python tools/bench_gpt.py 2>&1 | tail -2; done > gpurun_out/r06_gpt.txt 2>&1
for w in 128 64; do echo "== DTTS_STREAM_TOKEN_WGS=$w"; DTTS_STREAM_TOKEN_WGS=$w python tools/pipeline_trace.py --requests 8 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r06_pipeline_trace.txt 2>&1
for w in 64 32; do DTTS_GPT_TOKEN_WGS=$w DTTS_GPT_TOKEN_TRACE=300 python tools/bench_gpt.py 2>&1 >/dev/null | grep -A12 "workgroup 0 of"; done > gpurun_out/r06_gpt_token_trace.txt 2>&1
# what is left of stage A's cost: the token kernel exiting at once (garbage results; the codes' count is fixed by suppress_eos)
for rep in 1 2; do for cfg in base exit; do
  case $cfg in base) E="DTTS_X=0";; exit) E="DTTS_GPT_TOKEN_ABLATE=4";; esac
  env $E DTTS_BENCH_NO_EXTRA=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d.get('power') or {}
print('$cfg rep=$rep: %.2f ms/step, %s W, %s MHz, %s J/step' % (d['ms_per_step'], p.get('mean_W'), p.get('mean_sclk_MHz'), p.get('energy_J_per_step')))"
done; done > gpurun_out/r06_stage_a_floor.txt 2>&1
tail -c 600 gpurun_out/r06_default_bench.json
