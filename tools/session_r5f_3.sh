set -x
(time timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -5)
REPS=2 tools/batch1_ab.sh "X=0" "DTTS_CONV_STAGES4_MAXWG_ALONE=0"
DTTS_PROF_SHAPES=1 BB=1 timeout 300 python tools/bench_forward.py 2>&1 | tail -12
