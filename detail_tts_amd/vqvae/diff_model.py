"""DiffusionTts mirror (reference: vqvae/diff_model.py:133-322) — the device work is libdetail_hip.so."""
from __future__ import annotations

import torch


class DiffusionTts:
    def __init__(self, rt, cfg):
        self.rt = rt
        self.model_channels = cfg["model_channels"]
        self.in_channels = cfg["in_channels"]
        self.out_channels = cfg["out_channels"]
        self.num_heads = cfg["num_heads"]

    def get_conditioning(self, conditioning_input, lengths=None):
        """vqvae/diff_model.py:221-229: mel [B,128,T] -> [B,1536]"""
        return self.rt.diff_conditioning(conditioning_input.float().contiguous(), lengths)

    def timestep_independent(self, aligned_conditioning, conditioning_latent, expected_seq_len, return_code_pred=False, lengths=None):
        """vqvae/diff_model.py:231-260 (latent branch): [B,n,768] -> [B,768,4n]"""
        if return_code_pred:
            raise NotImplementedError("return_code_pred is a training-only branch")
        if aligned_conditioning.dtype != torch.float32:
            raise NotImplementedError("token (code_converter) conditioning is not on the inference path")
        lat_cm = aligned_conditioning.permute(0, 2, 1).contiguous()
        out = self.rt.diff_timestep_independent(lat_cm, conditioning_latent.contiguous(), lengths)
        if expected_seq_len != out.shape[-1]:
            raise ValueError("expected_seq_len must be 4 * n (F.interpolate nearest x4 on the inference path)")
        return out

    def forward(self, x, timesteps, aligned_conditioning=None, conditioning_latent=None, precomputed_aligned_embeddings=None,
                conditioning_free=False, return_code_pred=False, lengths=None):
        """vqvae/diff_model.py:262-322.  `timesteps` are the model-side (0..3999) integer timesteps of the 50-step
        schedule, as _WrappedModel passes them (vqvae/utils/diffusion.py:1282-1287)."""
        if precomputed_aligned_embeddings is None and not conditioning_free:
            precomputed_aligned_embeddings = self.timestep_independent(aligned_conditioning, conditioning_latent, x.shape[-1])
        ts = set(int(t) for t in torch.as_tensor(timesteps).reshape(-1).tolist())
        if len(ts) != 1:
            raise ValueError("all batch rows must share one timestep (as in p_sample_loop)")
        t = ts.pop()
        tmap = self.rt.timestep_map
        if t not in tmap:
            raise ValueError(f"timestep {t} is not one of the {len(tmap)} sampling timesteps")
        return self.rt.diff_forward(x.float().contiguous(), tmap.index(t), None if conditioning_free else precomputed_aligned_embeddings,
                                    cond_free=conditioning_free, lens=lengths)

    __call__ = forward
