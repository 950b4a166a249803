cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_diffusion.py tests/test_gpu_fullsize.py -m gpu -x -q -k "not free_sampling" 2>&1 | tail -3
export DTTS_PROF_SHAPES=1
for i in 1 2; do
echo "== vec on";  python tools/bench_layer.py 2>&1 | grep -v "^$\|amdgpu.ids"
echo "== vec off"; DTTS_X3_EPI_VEC=0 python tools/bench_layer.py 2>&1 | grep -v "^$\|amdgpu.ids"
done
echo "== BB=16 on"; BB=16 python tools/bench_layer.py | tail -9
echo "== BB=16 off"; DTTS_X3_EPI_VEC=0 BB=16 python tools/bench_layer.py | tail -9
echo "== BB=1 on"; BB=1 python tools/bench_layer.py | tail -9
echo "== BB=1 off"; DTTS_X3_EPI_VEC=0 BB=1 python tools/bench_layer.py | tail -9
