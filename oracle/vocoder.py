"""Oracle for stage C: flow-VAE front + HiFiGAN generator (TEST INFRASTRUCTURE).

Restates vqvae/model_24k.py:848-863 (infer_flowvae), :71-124 (SpecEncoder),
:127-169 (ResidualCouplingBlock), :221-288 (Generator),
vqvae/modules/attentions.py:73-107,161-303,317-363 (Encoder/MHA/FFN) and
vqvae/modules/modules.py:152-229,240-334,393-475 (WN, ResBlock1, Flip, coupling).
"""
from __future__ import annotations

import math

import numpy as np

from . import ops, philox
from .gpt import mel_style_encoder

F32 = np.float32
LRELU_SLOPE = 0.1


def vits_attention(P, p, x, mask, n_heads=4, window=4):
    """attentions.MultiHeadAttention.forward (self-attention, window_size=4, heads_share).
    x [B,C,T]; mask [B,T] bool. Relative terms restated as a banded +-window sum."""
    B, C, T = x.shape
    dk = C // n_heads
    q = ops.conv1d(x, P[p + ".conv_q.weight"], P[p + ".conv_q.bias"]).reshape(B, n_heads, dk, T).transpose(0, 1, 3, 2)
    k = ops.conv1d(x, P[p + ".conv_k.weight"], P[p + ".conv_k.bias"]).reshape(B, n_heads, dk, T).transpose(0, 1, 3, 2)
    v = ops.conv1d(x, P[p + ".conv_v.weight"], P[p + ".conv_v.bias"]).reshape(B, n_heads, dk, T).transpose(0, 1, 3, 2)
    qs = (q / F32(math.sqrt(dk))).astype(F32)
    scores = np.einsum("bhid,bhjd->bhij", qs, k).astype(F32)
    Ek, Ev = P[p + ".emb_rel_k"][0], P[p + ".emb_rel_v"][0]              # [2w+1, dk]
    rel_logits = np.einsum("bhid,rd->bhir", qs, Ek).astype(F32)          # r = (j-i)+w
    ii = np.arange(T)
    for r in range(2 * window + 1):
        j = ii + r - window
        ok = (j >= 0) & (j < T)
        scores[:, :, ii[ok], j[ok]] += rel_logits[:, :, ii[ok], r]
    am = mask[:, None, :, None] & mask[:, None, None, :]
    scores = np.where(am, scores, F32(-1e4))
    pa = ops.softmax(scores, -1)
    out = np.einsum("bhij,bhjd->bhid", pa, v).astype(F32)
    for r in range(2 * window + 1):
        j = ii + r - window
        ok = (j >= 0) & (j < T)
        out[:, :, ii[ok]] += pa[:, :, ii[ok], j[ok]][..., None] * Ev[r][None, None, None, :]
    out = out.transpose(0, 1, 3, 2).reshape(B, C, T)
    return ops.conv1d(out, P[p + ".conv_o.weight"], P[p + ".conv_o.bias"])


def vits_ffn(P, p, x, maskf, k=3):
    """attentions.FFN.forward (relu, same padding)."""
    pl, pr = (k - 1) // 2, k // 2
    h = np.pad(x * maskf, ((0, 0), (0, 0), (pl, pr)))
    h = np.maximum(ops.conv1d(h, P[p + ".conv_1.weight"], P[p + ".conv_1.bias"]), 0)
    h = np.pad(h * maskf, ((0, 0), (0, 0), (pl, pr)))
    return ops.conv1d(h, P[p + ".conv_2.weight"], P[p + ".conv_2.bias"]) * maskf


def spec_encoder(P, x, lengths, n_layers=3):
    """SpecEncoder.forward (sample=True, g=None), vqvae/model_24k.py:111-124 -> (y, m, logs)."""
    B, C, T = x.shape
    mask = ops.sequence_mask(lengths, T)
    mf = mask[:, None, :].astype(F32)
    h = x * mf
    h = h * mf                                                            # Encoder.forward: x = x * x_mask
    for i in range(n_layers):
        y = vits_attention(P, f"enc_p.encoder.attn_layers.{i}", h, mask)
        h = ops.layer_norm_channels(h + y, P[f"enc_p.encoder.norm_layers_1.{i}.gamma"], P[f"enc_p.encoder.norm_layers_1.{i}.beta"])
        y = vits_ffn(P, f"enc_p.encoder.ffn_layers.{i}", h, mf)
        h = ops.layer_norm_channels(h + y, P[f"enc_p.encoder.norm_layers_2.{i}.gamma"], P[f"enc_p.encoder.norm_layers_2.{i}.beta"])
    h = h * mf
    y = ops.conv1d(h, P["enc_p.out_proj.weight"], P["enc_p.out_proj.bias"])
    stats = ops.conv1d(y, P["enc_p.proj.weight"], P["enc_p.proj.bias"]) * mf
    return y, stats[:, :C], stats[:, C:]


def wn(P, p, x, mf, g, hidden=192, n_layers=4):
    """modules.WN.forward, vqvae/modules/modules.py:204-229."""
    out = np.zeros_like(x)
    G = ops.conv1d(g, P[p + ".cond_layer.weight"], P[p + ".cond_layer.bias"])       # [B, 2*h*L, 1]
    for i in range(n_layers):
        a = ops.conv1d(x, P[p + f".in_layers.{i}.weight"], P[p + f".in_layers.{i}.bias"], padding=2)
        a = a + G[:, i * 2 * hidden:(i + 1) * 2 * hidden]
        acts = (np.tanh(a[:, :hidden].astype(np.float64)) * ops.sigmoid(a[:, hidden:])).astype(F32)
        rs = ops.conv1d(acts, P[p + f".res_skip_layers.{i}.weight"], P[p + f".res_skip_layers.{i}.bias"])
        if i < n_layers - 1:
            x = (x + rs[:, :hidden]) * mf
            out = out + rs[:, hidden:]
        else:
            out = out + rs
    return out * mf


def flow_reverse(P, z, mf, g, n_flows=4):
    """ResidualCouplingBlock.forward(reverse=True), vqvae/model_24k.py:166-169."""
    half = z.shape[1] // 2
    x = z
    for f in reversed(range(2 * n_flows)):
        if f % 2 == 1:
            x = x[:, ::-1]                                                # Flip
            continue
        p = f"flow.flows.{f}"
        x0, x1 = x[:, :half], x[:, half:]
        h = ops.conv1d(x0, P[p + ".pre.weight"], P[p + ".pre.bias"]) * mf
        h = wn(P, p + ".enc", h, mf, g)
        m = ops.conv1d(h, P[p + ".post.weight"], P[p + ".post.bias"]) * mf
        x1 = (x1 - m) * mf                                                # logs == 0 (mean_only)
        x = np.concatenate([x0, x1], 1)
    return np.ascontiguousarray(x, F32)


def resblock1(P, p, x, k, dilations=(1, 3, 5)):
    """modules.ResBlock1.forward (x_mask=None), vqvae/modules/modules.py:315-328."""
    for l, d in enumerate(dilations):
        xt = ops.leaky_relu(x, LRELU_SLOPE)
        xt = ops.conv1d(xt, P[p + f".convs1.{l}.weight"], P[p + f".convs1.{l}.bias"], padding=(k * d - d) // 2, dilation=d)
        xt = ops.leaky_relu(xt, LRELU_SLOPE)
        xt = ops.conv1d(xt, P[p + f".convs2.{l}.weight"], P[p + f".convs2.{l}.bias"], padding=(k - 1) // 2)
        x = xt + x
    return x


def generator(P, z, g, rates=(8, 4, 2, 2, 2), kernels=(16, 8, 2, 2, 2), rb_kernels=(3, 7, 11)):
    """Generator.forward, vqvae/model_24k.py:269-288."""
    x = ops.conv1d(z, P["dec.conv_pre.weight"], P["dec.conv_pre.bias"], padding=3)
    x = x + ops.conv1d(g, P["dec.cond.weight"], P["dec.cond.bias"])
    nk = len(rb_kernels)
    for i, (u, k) in enumerate(zip(rates, kernels)):
        x = ops.leaky_relu(x, LRELU_SLOPE)
        x = ops.conv_transpose1d(x, P[f"dec.ups.{i}.weight"], P[f"dec.ups.{i}.bias"], stride=u, padding=(k - u) // 2)
        xs = None
        for j, rk in enumerate(rb_kernels):
            r = resblock1(P, f"dec.resblocks.{i * nk + j}", x, rk)
            xs = r if xs is None else xs + r
        x = xs / F32(nk)
    x = ops.leaky_relu(x, 0.01)                                            # F.leaky_relu default slope
    x = ops.conv1d(x, P["dec.conv_post.weight"], None, padding=3)
    return np.tanh(x).astype(F32)


def infer_flowvae(P, mel, lengths, seed, sample_ids, noise_scale=0.667, noise=None, trace=None):
    """SynthesizerTrn.infer_flowvae, vqvae/model_24k.py:848-863, batched with per-sample
    lengths (the reference is batch-1, all-ones masks). mel [B,128,T] -> wav [B,1,256*T]."""
    B, _, T = mel.shape
    assert T % 4 == 0
    mask = ops.sequence_mask(lengths, T)
    mf = mask[:, None, :].astype(F32)
    g = mel_style_encoder(P, "ref_enc", mel * mf, lengths)
    x = ops.conv1d(mel, P["in_proj.weight"], P["in_proj.bias"], padding=1)
    _, m_p, logs_p = spec_encoder(P, x, lengths)
    if noise is None:
        noise = np.stack([philox.normal(seed, s, philox.STAGE_FLOW_PRIOR, 0, m_p.shape[1] * T).reshape(-1, T) for s in sample_ids])
    z_p = (m_p + noise * np.exp(logs_p) * F32(noise_scale)).astype(F32)
    z = flow_reverse(P, z_p, mf, g)
    o = generator(P, z, g)
    if trace is not None:
        trace.update(g=g, m_p=m_p, logs_p=logs_p, z_p=z_p, z=z)
    return o
