import os, sys, time, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from detail_tts_amd.vqvae.model_24k import SynthesizerTrn
from detail_tts_amd.weights import select_inference_params, synthetic_state_dict
W = select_inference_params(synthetic_state_dict(0))
model = SynthesizerTrn(W, folded=True)
B = 8
rsr = np.random.RandomState(91)
rn = [234, 180, 201, 97, 234, 234, 150, 222]; rrl = [936, 700, 936, 512, 801, 936, 936, 640]; rtl = [61, 40, 61, 25, 50, 61, 61, 33]
rrefer = torch.from_numpy((rsr.randn(B, 128, 936) * 2 - 5).astype(np.float32)).cuda()
rtext = np.zeros((B, 61), np.int32)
for b in range(B): rtext[b, : rtl[b] - 1] = rsr.randint(3, 255, rtl[b] - 1)
rcodes = [rsr.randint(0, 8192, size=rn[b]) for b in range(B)]
rreq = dict(text=torch.from_numpy(rtext), text_length=torch.tensor(rtl), refer=rrefer, refer_lengths=torch.tensor(rrl), sample_ids=list(range(B)), forced_codes=rcodes)
import hashlib
for cols in (0, 1, 0, 1):
    model.rt.set_option("conv_cols", cols)
    list(model.infer_stream((dict(rreq, seed=4300 + i) for i in range(2)), max_generate_length=235))
    torch.cuda.synchronize(); t = time.perf_counter()
    n = 6
    out = list(model.infer_stream((dict(rreq, seed=4310 + i) for i in range(n)), max_generate_length=235))
    torch.cuda.synchronize(); ms = (time.perf_counter() - t) / n * 1e3
    print(f"ragged batch, conv_cols = {cols}: {ms:.1f} ms per step, {sum(rn) * 1024 / 24000.0 / (ms * 1e-3):.1f} audio-s/s, wav sha {hashlib.sha256(out[-1][0].cpu().numpy().tobytes()).hexdigest()[:12]}")
