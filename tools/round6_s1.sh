#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/s1_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s1_pytest.log
tail -5 gpurun_out/s1_pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/s1_bench.json 2> gpurun_out/s1_bench.err; echo "bench rc=$?"
tail -c 1500 gpurun_out/s1_bench.json
timeout 900 bash tools/sweep_ablate.sh 2 > gpurun_out/s1_sweep_ablate.txt 2>&1
cat gpurun_out/s1_sweep_ablate.txt
timeout 1200 python tools/soak.py > gpurun_out/s1_soak.txt 2> gpurun_out/s1_soak.err; echo "soak rc=$?"
tail -8 gpurun_out/s1_soak.txt
