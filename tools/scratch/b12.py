import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from detail_tts_amd.vqvae.model_24k import SynthesizerTrn
from detail_tts_amd.weights import select_inference_params, synthetic_state_dict
m = SynthesizerTrn(select_inference_params(synthetic_state_dict(0)), folded=True)
rs = np.random.RandomState(2)
B = 12
refer = torch.from_numpy((rs.randn(B, 128, 200) * 2 - 5).astype(np.float32)).cuda()
text = torch.from_numpy(np.concatenate([rs.randint(3, 255, (B, 20)), np.zeros((B, 1), np.int64)], 1).astype(np.int32))
w12, l12 = m.infer(text, torch.full((B,), 21), refer, torch.full((B,), 200), batch=True, seed=5, sample_ids=list(range(B)), max_generate_length=9, suppress_eos=True, return_lengths=True)
w4, l4 = m.infer(text[:4], torch.full((4,), 21), refer[:4], torch.full((4,), 200), batch=True, seed=5, sample_ids=list(range(4)), max_generate_length=9, suppress_eos=True, return_lengths=True)
d = float((w12[:4] - w4).abs().max())
print("B=12 (one-wave GEMV path) vs B=4 (workgroup GEMV path): max |diff| of the first 4 waveforms", d, "rms", float(w4.pow(2).mean().sqrt()))
assert d < 1e-3
