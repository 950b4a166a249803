# interleaved A/B of the number of CFG chunks / streams (DTTS_CFG_STREAMS) on the default bench
for i in 1 2; do for m in 1 2 4; do DTTS_CFG_STREAMS=$m python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('cfg_streams', $m, d['value'], d['ms_per_step'], d['stage_ms']['diff_sample'])"; done; done
