"""Micro-benchmarks of the MFMA kernels at the headline shapes (B=16 = cond|uncond of batch 8, T=936)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from detail_tts_amd.packing import pack_conv
from detail_tts_amd.runtime import Runtime

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3

rs = np.random.RandomState(0)
shapes = [(768, 768, 1, 0), (768, 768, 3, 1), (768, 2304, 1, 0), (128, 768, 3, 1), (768, 256, 3, 1)]
B, T = int(os.environ.get("BB", 16)), int(os.environ.get("TT", 936))
extra = {}
for i, (cin, cout, k, pad) in enumerate(shapes):
    w = (rs.randn(cout, cin, k) / np.sqrt(cin * k)).astype(np.float32)
    wp, bp = pack_conv(w, rs.randn(cout).astype(np.float32))
    extra[f"c{i}.wp"], extra[f"c{i}.bp"] = wp, bp
r = Runtime({}, parts=(), extra=extra)
for i, (cin, cout, k, pad) in enumerate(shapes):
    x = torch.randn(B, cin, T, device="cuda")
    for _ in range(3): r.op_conv1d(f"c{i}", x, cout, k, pad=pad)
    r.profile_enable(True)
    for _ in range(10): r.op_conv1d(f"c{i}", x, cout, k, pad=pad)
    for st in r.profile_report():
        print(f"conv {cin:4d}->{cout:4d} k{k} B{B} T{T}: {st['name']:32s} {st['total_ms']/st['launches']*1e3:8.1f} us  {st['flops']/st['total_ms']/1e9:6.1f} TFLOP/s")
    r.profile_enable(False)
if os.environ.get("ATTN", "1") == "1":
    from detail_tts_amd.weights import select_inference_params, synthetic_state_dict, fold_weight_norm
    W = fold_weight_norm(synthetic_state_dict(0, only_prefixes=["diffusion."]))
    rt = Runtime(W, folded=True, parts=("diffusion",))
    x = torch.randn(B, 768, T, device="cuda")
    for _ in range(3): rt.op_attention_block("diffusion.layers.3.attn", x)
    rt.profile_enable(True)
    for _ in range(10): rt.op_attention_block("diffusion.layers.3.attn", x)
    for st in rt.profile_report():
        print(f"attention block B{B} T{T}: {st['name']:32s} {st['total_ms']/st['launches']*1e3:8.1f} us  {st['flops']/st['total_ms']/1e9:6.1f} TFLOP/s")
