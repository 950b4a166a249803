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
        """gpt/model.py:514-545.  hf_generate_kwargs understood: do_sample, top_p, top_k, temperature, repetition_penalty, length_penalty
        (inert without beams).  Returns LongTensor [B * num_return_sequences, <=max] incl. the stop token (and input_tokens in front).

        Off the `infer` path but part of the reference's surface: `do_sample=False` (HF greedy search: only the repetition penalty is a
        logits processor there, argmax), `num_return_sequences = n` (HF repeats every row n times - repeat_interleave - and samples the
        copies independently: `sample_ids` of length B * n name their noise streams, or of length B: copy r of row b draws from
        sample_ids[b] + r - the expanded ids must be distinct, else ValueError), `input_tokens [rows, k]` (mel tokens in front of the generated ones; with num_return_sequences = n > 1 the
        reference tiles them AND lets HF expand the batch again: n * n rows of the single prompt, reproduced), `typical_sampling`
        (HF TypicalLogitsWarper(mass=typical_mass), which HF applies between the repetition penalty and the temperature)."""
        nrs = int(num_return_sequences)
        assert nrs >= 1
        do_sample = bool(hf_generate_kwargs.get("do_sample", True))
        refer = speech_conditioning_latent.float().contiguous()
        B = refer.shape[0]
        it = None
        if input_tokens is not None:
            it = torch.as_tensor(input_tokens).cpu().numpy().astype(np.int32)
            assert it.ndim == 2
            if nrs > 1:
                # gpt/model.py:533-537: the reference tiles the prompt's fake inputs n times and the prefixes n // rows times (its torch.cat
                # needs n * B == n rows: ONE prompt), then HF's generate expands every row n times again (repeat_interleave): n * n returned
                # rows, row r starting with input_tokens[(r // n) % rows] and sampling from its own stream
                assert nrs % it.shape[0] == 0, "The number of return sequences must be divisible by the number of input sequences"
                if B != 1:
                    raise ValueError("input_tokens with num_return_sequences > 1: the reference concatenates num_return_sequences * B rows of "
                                     "inputs with num_return_sequences rows of prefixes - a single prompt only")
                it = np.stack([it[(r // nrs) % it.shape[0]] for r in range(nrs * nrs)])
                nrs = nrs * nrs
        cl = None if cond_lengths is None else torch.as_tensor(cond_lengths).reshape(-1).tolist()
        texts = self._texts(text_inputs, text_lengths)
        ids = list(range(B * nrs)) if sample_ids is None else list(sample_ids)
        if nrs > 1:
            refer = refer.repeat_interleave(nrs, 0).contiguous()
            cl = None if cl is None else [v for v in cl for _ in range(nrs)]
            texts = [t for t in texts for _ in range(nrs)]
            if len(ids) == B:
                ids = [i + r for i in ids for r in range(nrs)]
        assert len(ids) == B * nrs, "sample_ids: one per returned sequence (or one per prompt)"
        if len(set(ids)) != len(ids):
            # HF samples all copies independently: two returned sequences on ONE Philox stream would be the same draw sequence
            # (per-prompt ids [0, 1] with n = 2 expand to [0, 1, 1, 2]: ADVICE r05)
            raise ValueError(f"sample_ids expand to {ids}: two returned sequences would share a noise stream - pass B * num_return_sequences "
                             "distinct ids (or per-prompt ids at least num_return_sequences apart)")
        G = self.max_mel_tokens - 1 if max_generate_length is None else int(max_generate_length)
        forced = None
        if it is not None:
            assert it.shape[0] == B * nrs and it.shape[1] <= G
            forced = [row for row in it]
        if do_sample:
            samp = dict(top_k=hf_generate_kwargs.get("top_k", 50), top_p=hf_generate_kwargs.get("top_p", 1.0),
                        temperature=hf_generate_kwargs.get("temperature", 1.0))
        else:       # greedy search = the one largest logit after the repetition penalty: top-k 1 keeps exactly it (any draw picks it)
            samp = dict(top_k=1, top_p=1.0, temperature=1.0)
        codes, ncodes, lat = self.rt.gpt_generate(
            refer, cl, texts, seed, ids, max_generate_length=G, repetition_penalty=hf_generate_kwargs.get("repetition_penalty", 1.0),
            suppress_eos=suppress_eos, forced_uniforms=forced_uniforms, forced_codes=forced, forced_fill=-1,
            typical_mass=float(typical_mass) if typical_sampling else 0.0, **samp)
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
