"""Oracle for the whole path: SynthesizerTrn.infer (TEST INFRASTRUCTURE).

Follows vqvae/model_24k.py:774-810.  The reference is hard-wired to batch 1
(:775-778); a batch here is *defined* as independent single-utterance runs, so
this function loops over samples and never lets one sample see another.
"""
from __future__ import annotations

import numpy as np

from . import diffusion as D
from . import gpt as G
from . import vocoder as V

F32 = np.float32


def infer_one(P, text, refer, seed, sample_id, sched=None, forced_codes=None, max_generate_length=600,
              top_k=50, suppress_eos=False, diffusion_steps=None, noise_scale=0.667, trace=None):
    """text [L] int (as api.py passes it, i.e. with its trailing 0), refer [128,T_ref] -> wav [256*4n]."""
    sched = sched or D.make_schedule()
    text = np.asarray(text, np.int64)[None]
    refer = np.asarray(refer, F32)[None]
    rl = np.array([refer.shape[2]])
    if forced_codes is None:
        codes = G.generate(P, refer, rl, text, seed, [sample_id], max_generate_length, top_k=top_k,
                           suppress_eos=suppress_eos)
    else:
        codes = np.asarray(forced_codes, np.int64)[None]
        codes = np.concatenate([codes, [[G.STOP_MEL]]], 1)
    codes = codes[:, :-1]                                                     # model_24k.py:795
    latent = G.latents_teacher_forced(P, refer, rl, text, codes)             # :796-799
    cond = D.get_conditioning(P, refer)                                       # :802
    mel = D.do_spectrogram_diffusion(P, sched, latent, cond, seed, [sample_id], n_steps=diffusion_steps)  # :803
    mel = D.denormalize_mel(mel)                                              # :804
    T = mel.shape[-1]
    wav = V.infer_flowvae(P, mel, np.array([T]), seed, [sample_id], noise_scale)   # :809
    if trace is not None:
        trace.update(codes=codes, latent=latent, cond=cond, mel=mel)
    return wav[0, 0]


def infer_batch(P, texts, refers, seed, sample_ids, **kw):
    return [infer_one(P, t, r, seed, s, **kw) for t, r, s in zip(texts, refers, sample_ids)]
