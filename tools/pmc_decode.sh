#!/bin/bash
# HBM-side traffic of the persistent decode token kernel (one launch = one token of 8 rows): FETCH_SIZE / WRITE_SIZE in separate passes.
#   gpurun -- 'bash tools/pmc_decode.sh r03'   ->  gpurun_out/<tag>_pmc_decode.txt
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  BB=8 timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/${TAG}_pmcdec_$c -o dec --output-format csv -- python $R/tools/bench_gpt.py > $OUT/${TAG}_pmcdec_$c.log 2>&1
done
python - <<PY
import csv, glob
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("/tmp/${TAG}_pmcdec_%s/**/*counter_collection.csv" % c, recursive=True)[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "gpt_token_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c]
    res[c] = (len(vals), sum(vals) / max(len(vals), 1))
n, fetch = res["FETCH_SIZE"]
_, write = res["WRITE_SIZE"]
# rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB; MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE tallies wide (16 B / lane) streaming
# reads at half their bytes -> 2 x FETCH for the weight / KV stream (the 16-byte exchange polls are wide reads too)
fetch_b, write_b = fetch * 1024.0, write * 1024.0
with open("$OUT/${TAG}_pmc_decode.txt", "w") as o:
    o.write("${TAG}: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- BB=8 python tools/bench_gpt.py; gpt_token_kernel, %d launches (one token of 8 rows each)\n" % n)
    o.write("  per launch: FETCH_SIZE %.1f MB -> 2 x FETCH (gfx950 correction for 16 B / lane reads) = %.1f MB;   WRITE_SIZE %.1f MB\n" % (fetch_b / 1e6, 2 * fetch_b / 1e6, write_b / 1e6))
    o.write("  algorithmic per token at this shape (bench.py decode_bytes_per_token): weights 308.3 MB + KV rows (mean over the 234 tokens) 89.0 MB = 397 MB read;\n")
    o.write("  exchange words: 5 hops x 10 layers: reads 128 workgroups x (3 x 32 KB all-gathers + 32 KB partials + qkv) ~ 170 MB from L2 / fabric, writes ~ 45 MB\n")
print(open("$OUT/${TAG}_pmc_decode.txt").read())
PY
