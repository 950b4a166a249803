"""Oracle for stage A: conditioning encoder, GPT-2 stack, sampler (TEST INFRASTRUCTURE).

Restates gpt/model.py (UnifiedVoice / GPT2InferenceModel), the MelStyleEncoder of
gpt/modules/modules.py:642-720, and the third-party HuggingFace arithmetic the
reference calls into: GPT2Model blocks and
GenerationMixin._sample logits processing.  `P` holds folded fp32
weights keyed by reference state-dict names.
"""
from __future__ import annotations

import math

import numpy as np

from . import ops, philox

F32 = np.float32

START_TEXT, STOP_TEXT = 255, 0
START_MEL, STOP_MEL = 8192, 8193


def mel_style_encoder(P, p, x, lengths=None):
    """MelStyleEncoder.forward, gpt|vqvae/modules/modules.py:696-720.
    x [B,n_mel,T] (caller applies any input masking), lengths [B] or None -> [B,out,1]."""
    B, _, T = x.shape
    pad = None if lengths is None else ~ops.sequence_mask(lengths, T)      # True = padded
    h = x.transpose(0, 2, 1)
    h = ops.mish(ops.linear(h, P[p + ".spectral.0.fc.weight"], P[p + ".spectral.0.fc.bias"]))
    h = ops.mish(ops.linear(h, P[p + ".spectral.3.fc.weight"], P[p + ".spectral.3.fc.bias"]))
    H = h.shape[-1]
    h = h.transpose(0, 2, 1)
    for i in range(2):
        u = ops.conv1d(h, P[p + f".temporal.{i}.conv1.conv.weight"], P[p + f".temporal.{i}.conv1.conv.bias"], padding=2)
        h = h + u[:, :H] * ops.sigmoid(u[:, H:])
    h = h.transpose(0, 2, 1)                                              # [B,T,H]
    if pad is not None:
        h = np.where(pad[:, :, None], F32(0), h)
    n_head = 2
    dk = H // n_head
    q = ops.linear(h, P[p + ".slf_attn.w_qs.weight"], P[p + ".slf_attn.w_qs.bias"]).reshape(B, T, n_head, dk)
    k = ops.linear(h, P[p + ".slf_attn.w_ks.weight"], P[p + ".slf_attn.w_ks.bias"]).reshape(B, T, n_head, dk)
    v = ops.linear(h, P[p + ".slf_attn.w_vs.weight"], P[p + ".slf_attn.w_vs.bias"]).reshape(B, T, n_head, dk)
    att = np.einsum("bihd,bjhd->bhij", q, k).astype(F32) / F32(np.power(H, 0.5))   # temperature sqrt(d_model)
    if pad is not None:
        att = np.where(pad[:, None, None, :], -np.inf, att)
    att = ops.softmax(att, -1)
    o = np.einsum("bhij,bjhd->bihd", att, v).astype(F32).reshape(B, T, H)
    h = ops.linear(o, P[p + ".slf_attn.fc.weight"], P[p + ".slf_attn.fc.bias"]) + h
    y = ops.linear(h, P[p + ".fc.fc.weight"], P[p + ".fc.fc.bias"])         # [B,T,out]
    if pad is None:
        w = y.mean(1)
    else:
        y = np.where(pad[:, :, None], F32(0), y)
        w = y.sum(1) / (~pad).sum(1)[:, None].astype(F32)
    return w[:, :, None].astype(F32)


def gpt2_block(P, l, x, causal_from=0, kv=None):
    """One HF GPT-2 block (pre-LN), 16 heads x 48, on x [B,L,768].
    kv: optional (k_past, v_past) each [B,H,Lp,48]; returns (y, (k_all, v_all))."""
    p = f"gpt.gpt.h.{l}"
    B, L, D = x.shape
    H = 16
    dh = D // H
    h = ops.layer_norm_last(x, P[p + ".ln_1.weight"], P[p + ".ln_1.bias"])
    qkv = (h @ P[p + ".attn.c_attn.weight"] + P[p + ".attn.c_attn.bias"]).astype(F32)
    q, k, v = [t.reshape(B, L, H, dh).transpose(0, 2, 1, 3) for t in np.split(qkv, 3, -1)]
    if kv is not None:
        k = np.concatenate([kv[0], k], 2)
        v = np.concatenate([kv[1], v], 2)
    Lk = k.shape[2]
    s = np.einsum("bhid,bhjd->bhij", q, k).astype(F32) / F32(math.sqrt(dh))
    qi = np.arange(Lk - L, Lk)[:, None]
    s = np.where(np.arange(Lk)[None, :] <= qi, s, -np.inf)
    a = np.einsum("bhij,bhjd->bhid", ops.softmax(s, -1), v).astype(F32)
    a = a.transpose(0, 2, 1, 3).reshape(B, L, D)
    x = x + (a @ P[p + ".attn.c_proj.weight"] + P[p + ".attn.c_proj.bias"]).astype(F32)
    h = ops.layer_norm_last(x, P[p + ".ln_2.weight"], P[p + ".ln_2.bias"])
    h = ops.gelu_new((h @ P[p + ".mlp.c_fc.weight"] + P[p + ".mlp.c_fc.bias"]).astype(F32))
    x = x + (h @ P[p + ".mlp.c_proj.weight"] + P[p + ".mlp.c_proj.bias"]).astype(F32)
    return x, (k, v)


def gpt2_stack(P, emb, kv=None, layers=10):
    """GPT2Model(inputs_embeds=emb) with wpe==0 (gpt/model.py:233-234) incl. ln_f."""
    new = []
    x = emb
    for l in range(layers):
        x, c = gpt2_block(P, l, x, kv=None if kv is None else kv[l])
        new.append(c)
    return ops.layer_norm_last(x, P["gpt.gpt.ln_f.weight"], P["gpt.gpt.ln_f.bias"]), new


def final_norm(P, h):
    return ops.layer_norm_last(h, P["gpt.final_norm.weight"], P["gpt.final_norm.bias"])


def mel_head(P, h):
    return ops.linear(h, P["gpt.mel_head.weight"], P["gpt.mel_head.bias"])


def text_prefix_ids(text):
    """inference_speech_tortoise:517-518 — text [B,L] (api.py already padded one 0):
    pad stop, prepend start -> [B, L+2]."""
    t = np.pad(np.asarray(text, np.int64), ((0, 0), (0, 1)), constant_values=STOP_TEXT)
    return np.pad(t, ((0, 0), (1, 0)), constant_values=START_TEXT)


def prefix_embeddings(P, refer, refer_lengths, text):
    """[cond ; text] prefix of gpt/model.py:517-526 -> [B, P, 768]."""
    ids = text_prefix_ids(text)
    temb = P["gpt.text_embedding.weight"][ids] + P["gpt.text_pos_embedding.emb.weight"][: ids.shape[1]][None]
    cond = mel_style_encoder(P, "gpt.conditioning_encoder", refer, refer_lengths).transpose(0, 2, 1)
    return np.concatenate([cond, temb], 1).astype(F32)


def mel_inputs_embeddings(P, mel_ids):
    """mel_embedding(ids)+mel_pos_embedding(0..k) for [start, c1..ck] (gpt/model.py:134-136)."""
    return (P["gpt.mel_embedding.weight"][mel_ids] + P["gpt.mel_pos_embedding.emb.weight"][: mel_ids.shape[1]][None]).astype(F32)


def logits_nocache(P, prefix, mel_ids):
    """GPT2InferenceModel.forward, kv_cache=False (gpt/model.py:107-185): full recompute.
    Returns (logits [B,L,8194], latent [B,L,768]) over the mel positions only."""
    emb = np.concatenate([prefix, mel_inputs_embeddings(P, mel_ids)], 1)
    h, _ = gpt2_stack(P, emb)
    lat = final_norm(P, h)[:, prefix.shape[1]:]
    return mel_head(P, lat), lat


def latents_teacher_forced(P, refer, refer_lengths, text, codes):
    """UnifiedVoice.forward(..., return_latent=True), gpt/model.py:429-491, as called at
    vqvae/model_24k.py:796-799 (wav_lengths = n*1024 so set_mel_padding is a no-op).
    text [B,L] (as passed to infer), codes [B,n] -> [B,n,768]."""
    B, n = codes.shape
    prefix = prefix_embeddings(P, refer, refer_lengths, text)
    mel = np.pad(np.asarray(codes, np.int64), ((0, 0), (0, 1)), constant_values=STOP_MEL)      # :464
    mel = np.pad(mel, ((0, 0), (1, 0)), constant_values=START_MEL)                               # :470
    emb = np.concatenate([prefix, mel_inputs_embeddings(P, mel)], 1)
    h, _ = gpt2_stack(P, emb)
    enc = final_norm(P, h[:, 1:])
    return enc[:, -(n + 2):][:, :-2]


# ----------------------------------------------------------------------------
# HF GenerationMixin._sample logits processing
# ----------------------------------------------------------------------------
def process_logits(scores, seen_ids, repetition_penalty=2.0, temperature=0.8, top_k=50, top_p=0.8, typical_mass=None):
    """scores [V] fp32 for one row; seen_ids: iterable of ids in the row's input_ids.
    Returns filtered scores (with -inf) ready for softmax.

    top_k defaults to 50: the reference passes no top_k (vqvae/model_24k.py:786-792), so HF's
    generation default applies — 50 in the 4.x GenerationConfig of the reference's era AND,
    measured here, still 50 in transformers 5.15.0 (generation/configuration_utils.py:617
    fills unset sampling params at generate() time; the golden `gpt_generate` fixture keeps
    38 of 8194 tokens at step 0, i.e. top-k 50 then top-p 0.8)."""
    s = np.array(scores, F32)
    ids = np.unique(np.asarray(list(seen_ids), np.int64))
    g = s[ids]
    s[ids] = np.where(g < 0, g * F32(repetition_penalty), g / F32(repetition_penalty))
    if typical_mass is not None and 0 < typical_mass < 1:
        # HF TypicalLogitsWarper (inference_speech_tortoise(typical_sampling=True), gpt/model.py:539): a custom processor, which HF
        # places between the repetition penalty and the sampling warpers (measured: tests/golden/make_golden_r5.py)
        lp = (s - (s.max() + np.log(np.sum(np.exp(s - s.max()), dtype=F32)))).astype(F32)
        pr = np.exp(lp)
        ent = -np.nansum(np.where(pr > 0, pr * lp, 0.0), dtype=F32)
        key = np.abs(-lp - ent).astype(F32)
        order = np.lexsort((np.arange(s.size), key))                      # ascending by key, then id
        cum = np.cumsum(pr[order], dtype=F32)
        last = min(int(np.sum(cum < F32(typical_mass))), s.size - 1)
        s = np.where(key > key[order][last], -np.inf, s).astype(F32)
    s = s / F32(temperature)
    if top_k:
        kth = np.sort(s)[-min(top_k, s.size)]
        s = np.where(s < kth, -np.inf, s).astype(F32)
    if top_p is not None and top_p < 1.0:
        order = np.argsort(s, kind="stable")                    # ascending
        cum = np.cumsum(ops.softmax(s[order]).astype(F32), dtype=F32)
        remove = cum <= F32(1 - top_p)
        remove[-1:] = False
        s[order[remove]] = -np.inf
    return s


def draw_token(filtered, u):
    """Inverse-CDF draw in vocabulary order (our multinomial spec): first id whose
    inclusive cumulative probability exceeds u*total."""
    p = ops.softmax(filtered).astype(np.float64)
    c = np.cumsum(p)
    return int(min(np.searchsorted(c, u * c[-1], side="right"), p.size - 1))


def generate(P, refer, refer_lengths, text, seed, sample_ids, max_generate_length=600, top_k=50,
             top_p=0.8, temperature=0.8, repetition_penalty=2.0, use_cache=True, forced_uniforms=None,
             suppress_eos=False, return_latents=False, input_tokens=None, do_sample=True, typical_mass=None):
    """UnifiedVoice.inference_speech_tortoise (gpt/model.py:514-545) + HF _sample.
    Returns codes [B, <=max] including the stop token (finished rows padded with 8193).
    input_tokens [B, k] (gpt/model.py:533-537): mel tokens in front of the generated ones - they are part of the returned codes and of the
    repetition penalty's history, positions 1 .. k; the draw at mel position p uses noise counter p.  do_sample=False: HF greedy search -
    only the repetition penalty is a logits PROCESSOR, temperature / top-k / top-p are sampling warpers and are not applied; argmax."""
    B = refer.shape[0]
    prefix = prefix_embeddings(P, refer, refer_lengths, text)
    Pn = prefix.shape[1]
    mel_ids = np.full((B, 1), START_MEL, np.int64)
    finished = np.zeros(B, bool)
    kv = None
    lat_all = []
    if use_cache:
        h, kv = gpt2_stack(P, np.concatenate([prefix, mel_inputs_embeddings(P, mel_ids)], 1))
        last = h[:, -1:]
    for step in range(max_generate_length):
        if use_cache:
            lat = final_norm(P, last)[:, 0]
        else:
            _, latfull = logits_nocache(P, prefix, mel_ids)
            lat = latfull[:, -1]
        logits = mel_head(P, lat)
        lat_all.append(lat)
        nxt = np.empty(B, np.int64)
        for b in range(B):
            sc = logits[b].copy()
            if suppress_eos:
                sc[STOP_MEL] = -np.inf
            seen = [1, START_MEL] + mel_ids[b, 1:].tolist()     # fake prefix ids (gpt/model.py:528-530)
            if input_tokens is not None and step < np.asarray(input_tokens).shape[1]:
                tok = int(np.asarray(input_tokens)[b, step])
            elif not do_sample:
                tok = int(np.argmax(process_logits(sc, seen, repetition_penalty, 1.0, None, 1.0, typical_mass)))
            else:
                f = process_logits(sc, seen, repetition_penalty, temperature, top_k, top_p, typical_mass)
                u = forced_uniforms[b][step] if forced_uniforms is not None else \
                    philox.uniform_scalar(seed, sample_ids[b], philox.STAGE_GPT_SAMPLE, step)
                tok = draw_token(f, u)
            nxt[b] = STOP_MEL if finished[b] else tok
        mel_ids = np.concatenate([mel_ids, nxt[:, None]], 1)
        finished |= nxt == STOP_MEL
        if finished.all():
            break
        if use_cache and step + 1 < max_generate_length:
            e = (P["gpt.mel_embedding.weight"][nxt] + P["gpt.mel_pos_embedding.emb.weight"][step + 1][None])[:, None].astype(F32)
            last, kv = gpt2_stack(P, e, kv=kv)
    codes = mel_ids[:, 1:]
    if return_latents:
        return codes, np.stack(lat_all, 1)
    return codes
