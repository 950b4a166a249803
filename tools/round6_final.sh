# the last GPU call of a round:  gpurun -- "bash tools/round6_final.sh"  - the full GPU suite, then everything DESIGN / profiles quote
set -x
mkdir -p gpurun_out
(time timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4) > gpurun_out/r06_final_pytest.log 2>&1
bash tools/round6_measure.sh > gpurun_out/r06_measure.log 2>&1
ROWS="1 8" bash tools/prof_gpt_rows.sh > gpurun_out/r06_gpt_rows.log 2>&1
tail -4 gpurun_out/r06_final_pytest.log
