cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
run() { echo "== $*"; env "$@" BB=2 DTTS_PROF_SHAPES=1 python tools/bench_layer.py 2>&1 | grep -v amdgpu.ids | grep -E "conv_x3|flash|gn_split|wall|profiled"; }
{
run DTTS_X=0
run DTTS_CONV_KSPLIT_K1=1
run DTTS_CONV_KSPLIT_K1=4
run DTTS_CONV_KSPLIT=2
run DTTS_CONV_KSPLIT=8
run DTTS_CONV_KSPLIT=1 DTTS_CONV_KSPLIT_K1=1
run DTTS_CONV_STAGES=4
run DTTS_CONV_STAGES=3
run DTTS_CONV_STAGES=2
run DTTS_CONV_KSPLIT_TILES=256 DTTS_CONV_KSPLIT_WGS=512
run DTTS_ATTN_KSPLIT=1
} 2>&1 | tee gpurun_out/b1_layer_sweep.txt
