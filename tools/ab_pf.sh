cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_diffusion.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do
echo "== L2PF on";  python tools/bench_layer.py 2>&1 | grep -v "^$"
echo "== L2PF off"; DTTS_CONV_L2PF=0 python tools/bench_layer.py 2>&1 | grep -v "^$"
done
echo "== BB=16 on"; BB=16 python tools/bench_layer.py | tail -12
echo "== BB=16 off"; DTTS_CONV_L2PF=0 BB=16 python tools/bench_layer.py | tail -12
echo "== BB=1 on"; BB=1 python tools/bench_layer.py | tail -12
echo "== BB=1 off"; DTTS_CONV_L2PF=0 BB=1 python tools/bench_layer.py | tail -12
