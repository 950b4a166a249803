# the last GPU call of a round:  gpurun -- "bash tools/round5_final.sh"  - the full GPU suite, then everything DESIGN / profiles quote (tools/round5_measure.sh), then the decode kernels by row count
set -x
(time timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4) > gpurun_out/r05_final_pytest.log 2>&1
bash tools/round5_measure.sh > gpurun_out/r05_measure.log 2>&1
ROWS="1 8" bash tools/prof_gpt_rows.sh > gpurun_out/r05_gpt_rows.log 2>&1
tail -4 gpurun_out/r05_final_pytest.log
