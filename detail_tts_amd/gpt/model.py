"""UnifiedVoice mirror (reference: gpt/model.py:265-545) — device work in libdetail_hip.so (dtts_gpt_*)."""
from __future__ import annotations

import numpy as np
import torch


class UnifiedVoice:
    def __init__(self, rt, cfg):
        self.rt = rt
        self.model_dim = cfg["model_dim"]
        self.max_mel_tokens = cfg["max_mel_tokens"]
        self.max_text_tokens = cfg["max_text_tokens"]
        self.start_text_token = cfg["start_text_token"]
        self.stop_text_token = 0
        self.start_mel_token = cfg["start_mel_token"]
        self.stop_mel_token = cfg["stop_mel_token"]
        self.mel_length_compression = cfg["mel_length_compression"]
        self.last_latents = None

    @staticmethod
    def _texts(text_inputs, text_lengths=None):
        t = torch.as_tensor(text_inputs).cpu().numpy().astype(np.int32)
        if t.ndim == 1:
            t = t[None]
        if text_lengths is None:
            return [r for r in t]
        tl = torch.as_tensor(text_lengths).reshape(-1).tolist()
        return [r[: int(n)] for r, n in zip(t, tl)]

    def inference_speech_tortoise(self, speech_conditioning_latent, cond_lengths, text_inputs, input_tokens=None,
                                  num_return_sequences=1, max_generate_length=None, typical_sampling=False, typical_mass=.9,
                                  text_lengths=None, seed=0, sample_ids=None, suppress_eos=False, forced_uniforms=None,
                                  **hf_generate_kwargs):
        """gpt/model.py:514-545.  hf_generate_kwargs understood: do_sample(True), top_p, top_k, temperature,
        repetition_penalty, length_penalty (inert without beams).  Returns LongTensor [B, <=max] incl. the stop token."""
        if input_tokens is not None or num_return_sequences != 1 or typical_sampling:
            raise NotImplementedError("input_tokens / num_return_sequences>1 / typical sampling are not on the infer path")
        if not hf_generate_kwargs.get("do_sample", True):
            raise NotImplementedError("greedy decoding is not on the infer path")
        refer = speech_conditioning_latent.float().contiguous()
        B = refer.shape[0]
        cl = None if cond_lengths is None else torch.as_tensor(cond_lengths).reshape(-1).tolist()
        G = self.max_mel_tokens - 1 if max_generate_length is None else int(max_generate_length)
        codes, ncodes, lat = self.rt.gpt_generate(
            refer, cl, self._texts(text_inputs, text_lengths), seed, list(range(B)) if sample_ids is None else sample_ids,
            max_generate_length=G, top_k=hf_generate_kwargs.get("top_k", 50), top_p=hf_generate_kwargs.get("top_p", 1.0),
            temperature=hf_generate_kwargs.get("temperature", 1.0), repetition_penalty=hf_generate_kwargs.get("repetition_penalty", 1.0),
            suppress_eos=suppress_eos, forced_uniforms=forced_uniforms)
        self.last_latents, self.last_ncodes = lat, ncodes
        n = int(ncodes.max())
        return torch.from_numpy(codes[:, :n].astype(np.int64)).to(refer.device)

    def forward(self, speech_conditioning_latent, cond_lengths, text_inputs, text_lengths, mel_codes, wav_lengths, types=None,
                text_first=True, raw_mels=None, return_attentions=False, return_latent=False, clip_inputs=False):
        """gpt/model.py:429-491, return_latent=True only (the inference use, vqvae/model_24k.py:796-799) -> [B, n, 768]"""
        if not return_latent or types is not None or raw_mels is not None or not text_first:
            raise NotImplementedError("only forward(..., return_latent=True) is on the inference path")
        refer = speech_conditioning_latent.float().contiguous()
        cl = None if cond_lengths is None else torch.as_tensor(cond_lengths).reshape(-1).tolist()
        codes = torch.as_tensor(mel_codes).cpu().numpy().astype(np.int32)
        wl = torch.as_tensor(wav_lengths).reshape(-1).tolist()
        code_list = []
        for b in range(codes.shape[0]):
            row = codes[b].copy()
            end = int(wl[b]) // self.mel_length_compression + 1            # set_mel_padding, gpt/model.py:377-390
            if end < row.shape[-1]:
                row[end:] = self.stop_mel_token
            code_list.append(row)
        lat = self.rt.gpt_latents(refer, cl, self._texts(text_inputs, text_lengths), code_list)
        return lat.permute(0, 2, 1).contiguous()

    __call__ = forward
