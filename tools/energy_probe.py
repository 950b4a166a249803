"""Board power, shader clock and energy per call of the trunk's kernels, each looped alone for ~1.5 s (hwmon sampler of bench.py), at the
bench's per-stream shape (8 samples, T = 936).  Under the bench stage B sits at the 1400 W power limit, so what a kernel costs the step
is its ENERGY, not its duration alone: this table ranks them.   python tools/energy_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import PowerSampler
from detail_tts_amd.packing import pack_conv
from detail_tts_amd.runtime import Runtime
from detail_tts_amd.weights import synthetic_state_dict, fold_weight_norm

B, T = int(os.environ.get("BB", 8)), int(os.environ.get("TT", 936))
SECS = float(os.environ.get("SECS", 1.5))
rs = np.random.RandomState(0)
shapes = {"conv k1 768->768": (768, 768, 1, 0), "conv k3 768->768": (768, 768, 3, 1), "conv k1 768->2304": (768, 2304, 1, 0)}
extra = {}
for i, (name, (cin, cout, k, pad)) in enumerate(shapes.items()):
    w = (rs.randn(cout, cin, k) / np.sqrt(cin * k)).astype(np.float32)
    extra[f"c{i}.wp"], extra[f"c{i}.bp"] = pack_conv(w, rs.randn(cout).astype(np.float32))
r = Runtime({}, parts=(), extra=extra)
W = fold_weight_norm(synthetic_state_dict(0, only_prefixes=["diffusion."]))
rt = Runtime(W, folded=True, parts=("diffusion",))
xs = float(os.environ.get("XSCALE", 1.0))
x768 = torch.randn(B, 768, T, device="cuda") * xs
work = {}
for i, (name, (cin, cout, k, pad)) in enumerate(shapes.items()):
    work[name + " (+ split pass)"] = (lambda i=i, cout=cout, k=k, pad=pad: r.op_conv1d(f"c{i}", x768, cout, k, pad=pad))
work["AttentionBlock (GN, qkv, attention, proj)"] = lambda: rt.op_attention_block("diffusion.layers.3.attn", x768)
work["ResBlock (GN, k1, GN, k3)"] = lambda: rt.op_resblock("diffusion.layers.3.resblk", x768, step=7)
a = torch.randn(64 << 20, device="cuda")
bb = torch.empty_like(a)
work["copy 256 MB (HBM stream)"] = lambda: bb.copy_(a)


def loop(fn, secs):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ps = PowerSampler(0).start()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < secs:
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        n += 50
    dt = time.perf_counter() - t0
    return n, dt, ps.stop()


time.sleep(1.0)
idle = PowerSampler(0).start()
time.sleep(1.0)
idle = idle.stop()
print(f"idle: {idle}")
print(f"{'workload':46s} {'us/call':>9s} {'W':>7s} {'MHz':>6s} {'mJ/call':>9s}")
for name, fn in work.items():
    n, dt, p = loop(fn, SECS)
    us = dt / n * 1e6
    print(f"{name:46s} {us:9.1f} {p['mean_W']:7.0f} {p['mean_sclk_MHz']:6d} {us * 1e-6 * p['mean_W'] * 1e3:9.2f}")
# two streams, as in the bench: ResBlock on one, AttentionBlock on the other
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def both():
    with torch.cuda.stream(s1):
        rt.op_resblock("diffusion.layers.3.resblk", x768, step=7)
    with torch.cuda.stream(s2):
        rt.op_attention_block("diffusion.layers.4.attn", x768)
n, dt, p = loop(both, SECS)
print(f"{'ResBlock || AttentionBlock on two streams':46s} {dt / n * 1e6:9.1f} {p['mean_W']:7.0f} {p['mean_sclk_MHz']:6d} {dt / n * p['mean_W'] * 1e3:9.2f}")
