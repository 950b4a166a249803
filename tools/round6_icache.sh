# instruction-cache counters of the token kernels (alone): is the 64-workgroup kernel's 62 KB layer loop hitting the 64 KB cache?
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for w in 128 64 32; do
  rm -rf /tmp/ic_$w
  DTTS_GPT_TOKEN_WGS=$w timeout 600 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d /tmp/ic_$w -o gpt --output-format csv -- python $R/tools/bench_gpt.py > /tmp/ic_$w.log 2>&1
  f=$(find /tmp/ic_$w -name '*counter_collection.csv' | head -1)
  python - $w $f <<'PY'
import csv, sys, collections
w, f = sys.argv[1:]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "gpt_token" not in k: continue
    k = k.split("(")[0][-40:]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    if r["Counter_Name"] == "SQC_ICACHE_REQ": n[k] += 1
for k, v in acc.items():
    d = max(n[k], 1)
    print(f"wgs={w} {k}: launches {n[k]}, per launch: " + ", ".join(f"{c} {x / d:.3e}" for c, x in sorted(v.items())), " miss rate %.1f %%" % (100 * v.get("SQC_ICACHE_MISSES", 0) / max(v.get("SQC_ICACHE_REQ", 1), 1)))
PY
done 2>&1 | tee $OUT/r06_icache.txt
