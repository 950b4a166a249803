cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_gpt.py tests/test_gpu_vocoder.py tests/test_gpu_frontend.py tests/test_gpu_diffusion.py -x -q 2>&1 | tail -3
{
for d in 0 1; do
  echo "== DTTS_CONV_SMALL_DEEP=$d"
  DTTS_CONV_SMALL_DEEP=$d BB=8 python tools/bench_gpt.py 2>&1 | grep "G=2"
  DTTS_CONV_SMALL_DEEP=$d BB=1 python tools/bench_gpt.py 2>&1 | grep "G=2"
  DTTS_CONV_SMALL_DEEP=$d python tools/bench_vocoder.py 2>&1 | grep -E "stage C|conv_gemm_kernel<64,64"
done
REPS=2 bash tools/batch1_ab.sh "DTTS_CONV_SMALL_DEEP=0" "DTTS_CONV_SMALL_DEEP=1"
} 2>&1 | tee gpurun_out/r06_conv_small_deep.txt
