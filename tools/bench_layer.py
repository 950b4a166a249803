"""Per-kernel timings of one diffusion layer (ResBlock + AttentionBlock) at the bench's per-stream shape (B=8, T=936)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from detail_tts_amd.runtime import Runtime
from detail_tts_amd.weights import synthetic_state_dict, fold_weight_norm

B, T = int(os.environ.get("BB", 8)), int(os.environ.get("TT", 936))
W = fold_weight_norm(synthetic_state_dict(0, only_prefixes=["diffusion."]))
rt = Runtime(W, folded=True, parts=("diffusion",))
# XSCALE: scale of the random input.  XSCALE=0 feeds zeros: every activation of the layer becomes a per-channel constant (the biases), the
# matrix pipe's operands barely toggle, and the kernels show what they do when power / clock is not the limit.
x = torch.randn(B, 768, T, device="cuda") * float(os.environ.get("XSCALE", 1.0))


def layer():
    y = rt.op_resblock("diffusion.layers.3.resblk", x, step=7)
    return rt.op_attention_block("diffusion.layers.3.attn", y)


for _ in range(3):
    layer()
rt.profile_enable(2)
for _ in range(20):
    layer()
tot = 0.0
for st in rt.profile_report():
    us = st["total_ms"] / st["launches"] * 1e3
    tot += st["total_ms"] / 20 * 1e3
    print(f"{st['name']:34s} {st['launches'] // 20:2d}/layer {us:8.1f} us  {st['flops'] / max(st['total_ms'], 1e-9) / 1e9:7.1f} TFLOP/s(eq)  {st['bytes'] / max(st['total_ms'], 1e-9) / 1e9:7.2f} TB/s(alg)")
print(f"profiled kernels per layer: {tot:.1f} us   (B={B}, T={T})")

# launch gaps: wall time of the same sequence without the profiler's events vs the sum of the kernel durations above
rt.profile_enable(False)
for _ in range(3):
    layer()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    layer()
e1.record()
torch.cuda.synchronize()
print(f"wall per layer without profiling: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us")
