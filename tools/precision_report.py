"""Numbers quoted in DESIGN.md: the split-precision trunk against the exact fp32-MFMA kernels and against the reference fixtures."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from detail_tts_amd.runtime import Runtime
from detail_tts_amd.weights import select_inference_params, synthetic_state_dict
from fullsize_inputs import inputs, sub
W = select_inference_params(synthetic_state_dict(0))
rt = Runtime(W, folded=True, parts=("diffusion",))
I = inputs()
G = dict(np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "fullsize.npz")))
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
out = {}
for flag in (1, 0):
    rt.set_option("conv_x3", flag)
    out[flag] = rt.diff_forward(dev(I["x"]), 47, dev(I["code_emb"])).cpu().numpy()[0]
rt.set_option("conv_x3", 1)
a, r = out[1], out[0]
print(f"full forward T=936: split-precision vs exact fp32 MFMA kernels: rel RMS {np.sqrt(np.mean((a - r) ** 2)) / np.sqrt(np.mean(r ** 2)):.2e}, max abs {np.abs(a - r).max():.2e} (|out| max {np.abs(r).max():.2f})")
for flag, nm in ((1, "split-precision"), (0, "exact fp32 MFMA")):
    s, t = sub(out[flag], G)
    print(f"  {nm:16s} vs the REFERENCE's own output (subsample): max abs {max(np.abs(s - G['fwd47_cond_s']).max(), np.abs(t - G['fwd47_cond_t']).max()):.2e}")
