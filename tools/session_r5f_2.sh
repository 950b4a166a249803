set -x
timeout 900 python -m pytest tests/test_gpu_gpt.py -x -q -m gpu -k "token or sixteen" 2>&1 | tail -3
for bb in 1 4; do BB=$bb timeout 300 python tools/bench_gpt.py 2>&1 | grep 'G='; done
R=$(pwd); mkdir -p gpurun_out
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/gptb1 && BB=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/gptb1 -o gpt -- python $R/tools/bench_gpt.py > /tmp/gptb1.log 2>&1; DB=$(find /tmp/gptb1 -name '*.db' | head -1); python $R/tools/rocprof_summary.py $DB $R/gpurun_out/gpt_rows_1.txt "decode kernels, 1 row"; head -14 $R/gpurun_out/gpt_rows_1.txt | cut -c1-170; python $R/tools/kernel_timeline.py $DB "" 2000 2>&1 | tail -30 )
REPS=2 tools/batch1_ab.sh "X=0" "DTTS_CONV_KSPLIT_WGS=400 DTTS_CONV_KSPLIT=6" "DTTS_CONV_KSPLIT_MAXTILE=192 DTTS_CONV_KSPLIT_WGS=400" "DTTS_CONV_KSPLIT_MAXTILE=192 DTTS_CONV_KSPLIT_WGS=400 DTTS_CONV_KSPLIT=6 DTTS_CONV_KSPLIT_K1=4" "DTTS_CONV_STAGES=4"
