"""Per-kernel timings of one DiffusionTts.forward pair (cond | uncond) at the bench's per-stream shape: B = 4 -> one 8-sample chunk,
T = 936 (BB / TT override).  DTTS_PROF_SHAPES=1 names the conv shapes; DTTS_GN_FUSE=0/1 selects the GroupNorm path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DTTS_PROF_SHAPES", "1")
import torch
from detail_tts_amd.runtime import Runtime
from detail_tts_amd.weights import synthetic_state_dict, fold_weight_norm

B, T = int(os.environ.get("BB", 4)), int(os.environ.get("TT", 936))
W = fold_weight_norm(synthetic_state_dict(0, only_prefixes=["diffusion."]))
rt = Runtime(W, folded=True, parts=("diffusion",))
x = torch.randn(B, 128, T, device="cuda")
ce = torch.randn(B, 768, T, device="cuda") * 0.5
N = 10


def fwd():
    return rt.diff_p_sample(x, ce, 25, 1, list(range(B)))


for _ in range(3):
    fwd()
rt.profile_enable(2)
for _ in range(N):
    fwd()
tot = 0.0
for st in sorted(rt.profile_report(), key=lambda s: -s["total_ms"]):
    us = st["total_ms"] / st["launches"] * 1e3
    tot += st["total_ms"] / N * 1e3
    print(f"{st['name']:34s} {st['launches'] / N:6.1f}/fwd {us:8.1f} us  {st['total_ms'] / N * 1e3:9.1f} us/fwd {st['flops'] / max(st['total_ms'], 1e-9) / 1e9:7.1f} TF(eq)")
print(f"profiled kernels per forward pair: {tot:.1f} us   (B={B}, T={T})")
rt.profile_enable(False)
for _ in range(3):
    fwd()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    fwd()
e1.record()
torch.cuda.synchronize()
print(f"wall per forward pair without profiling: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
