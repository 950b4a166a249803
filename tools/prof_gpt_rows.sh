#!/bin/bash
# per-kernel durations of the decode step at 1, 8 and 16 rows (ROWS="1 8 16") (rocprofv3 kernel trace of tools/bench_gpt.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for bb in ${ROWS:-1 8 16}; do
  rm -rf /tmp/gptrows_$bb
  BB=$bb timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/gptrows_$bb -o gpt -- python $R/tools/bench_gpt.py > /tmp/gptrows_$bb.log 2>&1
  DB=$(find /tmp/gptrows_$bb -name '*.db' | head -1)
  python $R/tools/rocprof_summary.py $DB $OUT/gpt_rows_$bb.txt "decode kernels, $bb rows"
  head -16 $OUT/gpt_rows_$bb.txt | cut -c1-170
done
