cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_diffusion.py tests/test_gpu_fullsize.py -m gpu -x -q -k "not free_sampling" 2>&1 | tail -2
export DTTS_PROF_SHAPES=1
for bb in 8 16 1; do
echo "== BB=$bb default"; BB=$bb python tools/bench_layer.py 2>&1 | grep "conv_x3\|wall"
echo "== BB=$bb stages=3 for all"; DTTS_CONV_STAGES=3 BB=$bb python tools/bench_layer.py 2>&1 | grep "conv_x3\|wall"
done
echo "== BB=8 stages=2 (old loop) for all"; DTTS_CONV_STAGES=2 python tools/bench_layer.py 2>&1 | grep "conv_x3\|wall"
