# interleaved A/B on the default bench:  bash tools/ab_merge.sh "ENV=a" "ENV=b" ...
for i in 1 2; do for e in "$@"; do env $e python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$e', d['value'], d['ms_per_step'], d['stage_ms'])"; done; done
