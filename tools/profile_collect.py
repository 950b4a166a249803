#!/usr/bin/env python3
"""Turn gpurun_out/<tag>_* (tools/profile_round.sh) into the committed summaries under profiles/."""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import rocprof_summary

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go, pf = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")
for name in (f"{tag}_bench_kernel_stats.txt", f"{tag}_default_bench.json"):          # summarised on the box by tools/profile_round.sh
    src = os.path.join(go, name)
    if os.path.exists(src):
        txt = open(src).read()
        if name.endswith(".json"):
            txt = [l for l in txt.splitlines() if l.startswith("{")][-1] + "\n"
        open(os.path.join(pf, name), "w").write(txt)
line = [l for l in open(os.path.join(go, f"{tag}_bench.json")) if l.startswith("{")] if os.path.exists(os.path.join(go, f"{tag}_bench.json")) else []
if line:
    open(os.path.join(pf, f"{tag}_bench.json"), "w").write(line[-1])
# HBM-side traffic of the conv_x3 launches of one layer: dispatch order inside a layer is c1 (1x1), c2 (k3), qkv (1x1, M=2304), proj (1x1)
names = ["768->768 k1", "768->768 k3", "768->2304 k1", "768->768 k1 (proj)"]
res = collections.OrderedDict((n, {}) for n in names)
other = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(os.path.join(go, f"{tag}_pmc_{c}", "**", "*counter_collection.csv"), recursive=True)
    if not f:
        continue
    rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Dispatch_Id"]))
    conv = [r for r in rows if "conv_x3_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c]
    for i, n in enumerate(names):
        v = [float(r["Counter_Value"]) for r in conv[i::4]]
        res[n][c] = sum(v) / max(len(v), 1)
    for k in ("gn_split_planes", "split_planes_kernel", "flash_attn"):   # the other trunk kernels
        v = [float(r["Counter_Value"]) for r in rows if k in r["Kernel_Name"] and r["Counter_Name"] == c and ("gn_" in r["Kernel_Name"]) == k.startswith("gn_")]
        if v:
            other[k][c] = sum(v) / len(v)
B, T = 8, 936
out = collections.OrderedDict()
for n in names:
    cin, rest = n.split("->")
    cout, k = rest.split()[0], int(rest.split()[1][1:])
    cin, cout = int(cin), int(cout)
    if "FETCH_SIZE" not in res[n]:
        continue
    out[n] = {"fetch_MB": round(res[n]["FETCH_SIZE"] * 1024 / 1e6, 1), "write_MB": round(res[n].get("WRITE_SIZE", 0) * 1024 / 1e6, 1),
              "alg_in_MB": round(B * cin * T * 4 / 1e6, 1), "alg_w_MB": round(cin * cout * k * 4 / 1e6, 1),
              "alg_out_MB": round(B * cout * T * 4 * 2 / 1e6 if "2304" not in n and k == 3 or "proj" in n else B * cout * T * 4 / 1e6, 1)}
for k, v in other.items():
    out[k] = {"fetch_MB": round(v.get("FETCH_SIZE", 0) * 1024 / 1e6, 1), "write_MB": round(v.get("WRITE_SIZE", 0) * 1024 / 1e6, 1)}
if out:
    import hashlib
    # the counters belong to THIS kernel source: bench.py ignores the file once conv_x3.hip has changed
    out["conv_x3_sha1"] = hashlib.sha1(open(os.path.join(root, "detail_tts_amd", "csrc", "conv_x3.hip"), "rb").read()).hexdigest()
    json.dump(out, open(os.path.join(pf, f"{tag}_pmc_layer_traffic.json"), "w"), indent=1)
    with open(os.path.join(pf, f"{tag}_pmc_layer_traffic.txt"), "w") as fh:
        fh.write(f"{tag}: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python tools/bench_layer.py  (one diffusion layer, B=8, T=936:\n"
                 "the per-launch shape of the two-stream bench).  Units: KiB per dispatch (rocprofv3) -> MB = KiB*1024/1e6.  gfx950 caveat\n"
                 "(MI355X_MICROARCH.md, HBM section): FETCH_SIZE under-reports wide (16 B/lane) streaming reads by 2x; both conv_x3 operand\n"
                 "streams are 16 B/lane LDS-DMA loads, so the raw fetch value is a LOWER bound (bench.py doubles it).  alg_* = algorithmic bytes\n"
                 "(4 B per split-precision input element = two fp16 planes, 4 B per fp32 output element, residual read included where the conv adds one).\n\n")
        for k, v in out.items():
            if not isinstance(v, dict):
                fh.write(f"{k} = {v}\n")
                continue
            fh.write(f"{k:24s} " + "  ".join(f"{a}={b}" for a, b in v.items()) + "\n")
    print(json.dumps(out, indent=1))

# HBM-side traffic of the vocoder's fused ResBlock1 launches (tools/profile_round.sh: rocprofv3 --pmc on tools/bench_vocoder.py)
voc = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(os.path.join(go, f"{tag}_pmcvoc_{c}", "**", "*counter_collection.csv"), recursive=True)
    if not f:
        continue
    rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Dispatch_Id"]))
    fused = [r for r in rows if "resblock1x3_fused_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c]
    for i, nm in enumerate(("stage 4 (25 ch, T = 119808)", "stage 5 (12 ch, T = 239616)")):
        v = [float(r["Counter_Value"]) for r in fused[i::2]]
        if v:
            voc.setdefault(nm, {})[c] = sum(v) / len(v)
if voc:
    with open(os.path.join(pf, f"{tag}_pmc_vocoder.txt"), "w") as fh:
        fh.write(f"{tag}: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python tools/bench_vocoder.py (B = 8, T = 936):\n"
                 "resblock1x3_fused_kernel (the three ResBlock1 branches + mean of a narrow generator stage in one LDS-resident kernel).\n"
                 "FETCH_SIZE doubled (gfx950: 16 B/lane streaming reads are tallied at half).  Algorithmic bytes = x in + mean out = 8 B x C x T x B.\n\n")
        for nm, v in voc.items():
            C = 25 if "25 ch" in nm else 12
            Tn = 119808 if C == 25 else 239616
            alg = 8.0 * C * Tn * 8 / 1e6
            fe, wr = 2 * v.get("FETCH_SIZE", 0) * 1024 / 1e6, v.get("WRITE_SIZE", 0) * 1024 / 1e6
            fh.write(f"{nm:32s} fetch {fe:8.1f} MB  write {wr:8.1f} MB  algorithmic {alg:8.1f} MB  ratio {(fe + wr) / alg:5.2f}\n")
    print(open(os.path.join(pf, f"{tag}_pmc_vocoder.txt")).read())
