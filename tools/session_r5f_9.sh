set -x
for bb in 1 8; do BB=$bb timeout 300 python tools/bench_gpt.py 2>&1 | grep 'G='; done
timeout 300 python tools/prefill_prof.py 2>&1 | tail -12 | cut -c1-200
python tools/bench_vocoder.py 2>&1 | head -8 | cut -c1-200
(time timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4)
REPS=1 tools/batch1_ab.sh "X=0"
