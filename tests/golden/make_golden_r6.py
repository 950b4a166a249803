#!/usr/bin/env python3
"""Round-6 fixture FROM THE REFERENCE ITSELF (build container only, CPU): the empty-codes branch of SynthesizerTrn.infer_gpt
(vqvae/model_24k.py:833-834) - when the GPT emits the stop token first, `codes[:, :-1]` is empty, the reference substitutes a ZERO
latent of 16 frames (`torch.zeros(B, C, 16)`), adds g_vq, decodes 64 mel frames through vq_dec and vocodes them.  The reference's own
`infer_gpt` runs with `inference_speech_tortoise` returning the stop token alone; the prompt is the one of make_golden.py's `vq_path`
fixture.  Stores inputs and expected outputs only.

    python tests/golden/make_golden_r6.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import SEED_N, build_reference_model, install_shim, philox_rng, save   # noqa: E402


def main():
    install_shim()
    import torch
    torch.set_grad_enabled(False)
    m = build_reference_model()
    g = m.gpt
    refer = np.load(os.path.join(HERE, "vq_path.npz"))["refer"]
    refer_t, rl = torch.from_numpy(refer), torch.tensor([refer.shape[2]])
    text_t = torch.zeros((1, 4), dtype=torch.long)
    orig = g.inference_speech_tortoise
    g.inference_speech_tortoise = lambda *a, **k: torch.tensor([[g.stop_mel_token]])
    try:
        with philox_rng(sample_id=9):
            wav = m.infer_gpt(text_t, torch.tensor([4]), refer_t, rl)
    finally:
        g.inference_speech_tortoise = orig
    # the intermediate the branch feeds the vocoder with: vq_dec(zeros(1, 768, 16) + g_vq)
    import vqvae.modules.commons as commons
    rmask = commons.sequence_mask(rl, refer_t.size(2)).unsqueeze(1).float()
    g_vq = m.vq_ref_enc(refer_t * rmask, rmask)
    recon = m.vq_dec(torch.zeros(1, g_vq.shape[1], 16) + g_vq)
    save("infer_gpt_empty", refer=refer, recon=recon, wav=wav, seed=np.array(SEED_N), sample_id=np.array(9))


if __name__ == "__main__":
    main()
