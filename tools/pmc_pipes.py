#!/usr/bin/env python3
"""Summarise gpurun_out/<tag>_pmc_*.csv (tools/pmc_pipes.sh) per kernel: means per dispatch of every collected counter + derived
ratios (MFMA busy, instruction mix, wait split, L2 hit rate, corrected HBM traffic).   python tools/pmc_pipes.py <tag> [out.txt]"""
import collections, csv, glob, os, re, sys

tag = sys.argv[1]


def clean(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "").replace("dtts::", "")
    return re.sub(r"\(.*", "", n)


root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go = os.path.join(root, "gpurun_out")
val = collections.defaultdict(lambda: collections.defaultdict(list))     # kernel -> counter -> values
for f in sorted(glob.glob(os.path.join(go, f"{tag}_pmc_*.csv"))):
    for r in csv.DictReader(open(f)):
        k = clean(r["Kernel_Name"])
        val[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
kt = os.path.join(go, f"{tag}_kernel_trace.csv")
if os.path.exists(kt):
    for r in csv.DictReader(open(kt)):
        k = clean(r["Kernel_Name"])
        dur[k].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
bbf = os.path.join(go, f"{tag}_pmc_bb.txt")
bb = open(bbf).read().strip() if os.path.exists(bbf) else os.environ.get("BB", "?")      # written by tools/pmc_pipes.sh on the GPU box
lines = [f"{tag}: rocprofv3 --pmc <set> --kernel-trace -- python tools/bench_layer.py (one diffusion layer, B = {bb} samples per launch, T = 936);",
         "means per dispatch.  SQ_* cycle counters (other than MFMA_BUSY) are quad-cycles summed over waves; MFMA busy % is quoted against",
         "1024 SIMDs x duration x 2.4 GHz (a lower bound: the clock under load is lower).  FETCH_SIZE doubled (gfx950 16 B/lane correction).", ""]
for k, cs in sorted(val.items(), key=lambda kv: -sum(dur.get(kv[0], [0]))):
    if not any(s in k for s in ("conv_x3", "flash_attn", "gn_split", "split_planes")):
        continue
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    d = sum(dur[k]) / len(dur[k]) if dur.get(k) else float("nan")
    lines.append(f"{k[:60]:60s} dispatches {len(dur.get(k, []))}  avg {d:8.1f} us")
    g = m.get
    if g("SQ_VALU_MFMA_BUSY_CYCLES"):
        lines.append(f"    MFMA busy {100 * g('SQ_VALU_MFMA_BUSY_CYCLES') / (1024 * d * 1e-6 * 2.4e9):5.1f} %   insts: MFMA {g('SQ_INSTS_MFMA', 0):.3g} VALU {g('SQ_INSTS_VALU', 0):.3g} "
                     f"SALU {g('SQ_INSTS_SALU', 0):.3g} LDS {g('SQ_INSTS_LDS', 0):.3g} VMEM_RD {g('SQ_INSTS_VMEM_RD', 0):.3g} SMEM {g('SQ_INSTS_SMEM', 0):.3g}")
    if g("SQ_WAVE_CYCLES"):
        w = g("SQ_WAVE_CYCLES")
        lines.append(f"    wave-cycles: active {100 * g('SQ_ACTIVE_INST_ANY', 0) / w:5.1f} %  wait(waitcnt/barrier) {100 * g('SQ_WAIT_ANY', 0) / w:5.1f} %  issue-stall "
                     f"{100 * g('SQ_WAIT_INST_ANY', 0) / w:5.1f} %   waves {g('SQ_WAVES', 0):.0f}")
        if g("SQ_ACTIVE_INST_VALU") is not None:
            lines.append("    active split: " + "  ".join(f"{c[15:]} {100 * g(c, 0) / w:4.1f} %" for c in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM",
                                                                                                      "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC") if g(c) is not None)
                         + f"   LDS issue-stall {100 * g('SQ_WAIT_INST_LDS', 0) / w:4.1f} %")
    if g("SQ_LDS_IDX_ACTIVE"):
        lines.append(f"    LDS: active cycles {g('SQ_LDS_IDX_ACTIVE'):.3g}  bank-conflict cycles {g('SQ_LDS_BANK_CONFLICT', 0):.3g}")
    if g("TCC_HIT_sum") is not None:
        lines.append(f"    L2: hit rate {100 * g('TCC_HIT_sum') / max(g('TCC_HIT_sum') + g('TCC_MISS_sum', 0), 1):5.1f} %  requests {g('TCC_REQ_sum', 0):.3g}")
    if g("FETCH_SIZE") is not None:
        lines.append(f"    HBM side: fetch {2 * g('FETCH_SIZE') * 1024 / 1e6:7.1f} MB (2 x FETCH_SIZE)  write {g('WRITE_SIZE', 0) * 1024 / 1e6:7.1f} MB")
    other = {c: v for c, v in m.items() if c.startswith(("TCP_", "GRBM", "SQ_INSTS_VALU_MFMA_MOPS", "SQ_INST_CYCLES", "SQ_LDS_ADDR"))}
    if other:
        lines.append("    " + "  ".join(f"{c} {v:.3g}" for c, v in other.items()))
txt = "\n".join(lines) + "\n"
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt)
sys.stdout.write(txt)
