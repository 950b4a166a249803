cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_diffusion.py -x -q 2>&1 | tail -3
{
for T in 300 500; do for ks in 1 4; do echo "== TT=$T DTTS_ATTN_KSPLIT=$ks"; TT=$T BB=2 DTTS_ATTN_KSPLIT=$ks python tools/bench_layer.py 2>&1 | grep -E "flash|wall"; done; done
} 2>&1 | tee gpurun_out/s4_ksplit_short.txt
SETTINGS=0 REQUESTS=500 timeout 1200 python tools/soak.py > gpurun_out/s4_soak.txt 2> gpurun_out/s4_soak.err; echo "soak rc=$?"; tail -4 gpurun_out/s4_soak.txt
