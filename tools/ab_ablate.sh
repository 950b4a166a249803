cd $GRAFT_REPO_ROOT
export DTTS_PROF_SHAPES=1 DTTS_CONV_L2PF=0
for a in 0 1 2 4 8 16 24 3 7 15; do
echo "== ablate $a"; DTTS_X3_ABLATE=$a python tools/bench_layer.py 2>&1 | grep "conv_x3"
done
