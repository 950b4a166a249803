#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_diffusion.py tests/test_gpu_fullsize.py -q -m gpu -k "attention or forward" 2>&1 | tail -3
for a in 1024 0; do
  echo "DTTS_CONV_ABLATE=$a"
  DTTS_CONV_ABLATE=$a DTTS_PROF_SHAPES=1 python tools/bench_layer.py 2>&1 | grep "M=2304\|wall"
  cd /tmp; export TMPDIR=/tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    DTTS_CONV_ABLATE=$a BB=8 timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/q_$a_$c -o layer --output-format csv -- python $R/tools/bench_layer.py > /dev/null 2>&1
    python - <<PY
import csv, glob
f = glob.glob("/tmp/q_$a_$c/**/*counter_collection.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "conv_x3_kernel<2" in r["Kernel_Name"] and r["Counter_Name"] == "$c"]
v = [float(r["Counter_Value"]) for r in rows]
print("  qkv conv $c: mean %.1f MB over %d dispatches%s" % (sum(v) / len(v) * 1024 / 1e6 * (2 if "$c" == "FETCH_SIZE" else 1), len(v), " (x2 corrected)" if "$c" == "FETCH_SIZE" else ""))
PY
    rm -rf /tmp/q_$a_$c
  done
  cd $R
done
STEPS=8 WARMUP=3 bash tools/ab_env.sh "DTTS_CONV_ABLATE=1024" "DTTS_AB=grouped" 3
