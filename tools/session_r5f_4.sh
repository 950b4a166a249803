set -x
timeout 900 python -m pytest tests/test_gpu_gpt.py -x -q -m gpu 2>&1 | tail -3
R=$(pwd); mkdir -p gpurun_out
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/gptb1 && BB=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/gptb1 -o gpt -- python $R/tools/bench_gpt.py > /tmp/gptb1.log 2>&1; DB=$(find /tmp/gptb1 -name '*.db' | head -1); python $R/tools/rocprof_summary.py $DB $R/gpurun_out/gpt_rows_1.txt "decode kernels, 1 row"; head -8 $R/gpurun_out/gpt_rows_1.txt | cut -c1-170 )
for bb in 8 16; do DTTS_GPT_TOKEN_TRACE=60 BB=$bb timeout 300 python tools/bench_gpt.py 2>&1 | grep -A3 'workgroup 0' | cut -c1-330; done
REPS=2 tools/batch1_ab.sh "X=0" "DTTS_CONV_STAGES=4" "DTTS_CONV_STAGES=3" "DTTS_CONV_STAGES3_MAXWG=100000" "DTTS_CONV_STAGES4_MAXWG=100000 DTTS_CONV_STAGES4_MAXWG_K3=300"
