"""Sampler object mirroring the subset of the reference's SpacedDiffusion that inference uses
(vqvae/utils/diffusion.py:1181-1220 + GaussianDiffusion.p_sample_loop :654-742).  The arithmetic lives in
libdetail_hip.so (dtts_diff_sample); this class carries the schedule constants and the call surface."""
from __future__ import annotations

import numpy as np


def space_timesteps(num_timesteps, section_counts):
    """vqvae/utils/diffusion.py:1223-1272 (list-of-ints form)."""
    if isinstance(section_counts, int):
        section_counts = [section_counts]
    size_per, extra = num_timesteps // len(section_counts), num_timesteps % len(section_counts)
    start, out = 0, []
    for i, count in enumerate(section_counts):
        size = size_per + (1 if i < extra else 0)
        if size < count:
            raise ValueError(f"cannot divide section of {size} steps into {count}")
        stride = 1 if count <= 1 else (size - 1) / (count - 1)
        cur = 0.0
        for _ in range(count):
            out.append(start + round(cur))
            cur += stride
        start += size
    return set(out)


def get_named_beta_schedule(name, n):
    if name != "linear":
        raise NotImplementedError(name)
    scale = 1000 / n
    return np.linspace(scale * 0.0001, scale * 0.02, n, dtype=np.float64)


class SpacedDiffusion:
    def __init__(self, use_timesteps, betas, conditioning_free=True, conditioning_free_k=2.0, **_ignored):
        self.use_timesteps = set(use_timesteps)
        ac = np.cumprod(1.0 - np.asarray(betas, np.float64))
        last, nb, self.timestep_map = 1.0, [], []
        for i, a in enumerate(ac):
            if i in self.use_timesteps:
                nb.append(1 - a / last)
                last = a
                self.timestep_map.append(i)
        self.betas = np.array(nb)
        self.num_timesteps = len(nb)
        self.original_num_steps = len(betas)
        self.conditioning_free = conditioning_free
        self.conditioning_free_k = conditioning_free_k

    def p_sample_loop(self, model, shape, noise=None, model_kwargs=None, progress=False, seed=0, sample_ids=None, lens=None, **_):
        """model: a detail_tts_amd DiffusionTts; returns x_0 (normalised mel) [B,128,T]."""
        emb = (model_kwargs or {}).get("precomputed_aligned_embeddings")
        if emb is None:
            raise ValueError("precomputed_aligned_embeddings is required (as in do_spectrogram_diffusion)")
        B = shape[0]
        sample_ids = list(range(B)) if sample_ids is None else sample_ids
        return model.rt.diff_sample(emb, seed, sample_ids, lens=lens, x_init=noise, denorm=False)
