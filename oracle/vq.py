"""Oracle for the VQ decode path of `infer_gpt` (TEST INFRASTRUCTURE).

Restates vqvae/model_24k.py:811-847 (infer_gpt), :610-624 (vq_dec), vqvae/modules/core_vq.py:188-190, 298-301, 377-383
(EuclideanCodebook.dequantize, VectorQuantization.decode, ResidualVectorQuantization.decode).
"""
from __future__ import annotations

import numpy as np

from . import ops
from .gpt import mel_style_encoder
from . import vocoder as V

F32 = np.float32


EMPTY_FRAMES = 16       # vqvae/model_24k.py:833-834: an empty code sequence decodes to a ZERO latent of 16 frames


def quantizer_decode(P, codes):
    """codes [B,n] -> [B,768,n]: embed lookup (codebook_dim 8) -> project_out -> 'b n d -> b d n'.  n = 0 (the stop token came first):
    the reference's substitute, zeros [B,768,16] (vqvae/model_24k.py:833-834)."""
    if np.asarray(codes).shape[-1] == 0:
        return np.zeros((np.asarray(codes).shape[0], P["quantizer.vq.layers.0.project_out.weight"].shape[0], EMPTY_FRAMES), F32)
    q = P["quantizer.vq.layers.0._codebook.embed"][np.asarray(codes, np.int64)]                 # [B,n,8]
    q = ops.linear(q, P["quantizer.vq.layers.0.project_out.weight"], P["quantizer.vq.layers.0.project_out.bias"])
    return np.ascontiguousarray(q.transpose(0, 2, 1), F32)


def conv_transpose1d_op(x, w, b, stride, padding, output_padding):
    y = ops.conv_transpose1d(x, w, None, stride=stride, padding=0)                              # full length (T-1)*s + k
    L = (x.shape[2] - 1) * stride - 2 * padding + w.shape[2] + output_padding
    y = np.pad(y, ((0, 0), (0, 0), (0, max(0, padding + L - y.shape[2]))))[:, :, padding:padding + L]
    return (y + b[None, :, None]).astype(F32)


def vq_dec(P, x):
    """nn.Sequential vq_dec, vqvae/model_24k.py:610-624: LN(ch) -> ConvT(768->384,k3,s2,p1,op1) -> SiLU -> ConvT(384->192) -> SiLU -> Conv k3."""
    h = ops.layer_norm_channels(x, P["vq_dec.1.weight"], P["vq_dec.1.bias"])
    h = ops.silu(conv_transpose1d_op(h, P["vq_dec.3.weight"], P["vq_dec.3.bias"], 2, 1, 1))
    h = ops.silu(conv_transpose1d_op(h, P["vq_dec.5.weight"], P["vq_dec.5.bias"], 2, 1, 1))
    return ops.conv1d(h, P["vq_dec.7.weight"], P["vq_dec.7.bias"], padding=1)


def vq_decode_mel(P, codes, refer, refer_lengths):
    """codes [B,n] (stop token already dropped), refer [B,128,T_ref] -> recon mel [B,128,4n] (infer_gpt :828-845)."""
    T = refer.shape[2]
    mf = ops.sequence_mask(refer_lengths, T)[:, None, :].astype(F32)
    latent = quantizer_decode(P, codes)
    g_vq = mel_style_encoder(P, "vq_ref_enc", refer * mf, refer_lengths)
    return vq_dec(P, latent + g_vq)


def infer_gpt_from_codes(P, codes, refer, seed, sample_id, noise_scale=0.667):
    """One utterance: codes [n] (without stop), refer [128,T_ref] -> wav."""
    mel = vq_decode_mel(P, np.asarray(codes)[None], np.asarray(refer, F32)[None], [refer.shape[1]])
    return V.infer_flowvae(P, mel, [mel.shape[2]], seed, [sample_id], noise_scale)[0, 0]


# ---------------------------------------------------------------------------------------------------- encode side
def vq_enc(P, y):
    """nn.Sequential vq_enc, vqvae/model_24k.py:606-615: LN(mel ch) -> conv k3 s2 -> SiLU -> conv k3 s2 -> SiLU -> conv k3."""
    h = ops.layer_norm_channels(y, P["vq_enc.1.weight"], P["vq_enc.1.bias"])
    h = ops.silu(ops.conv1d(h, P["vq_enc.3.weight"], P["vq_enc.3.bias"], stride=2, padding=1))
    h = ops.silu(ops.conv1d(h, P["vq_enc.5.weight"], P["vq_enc.5.bias"], stride=2, padding=1))
    return ops.conv1d(h, P["vq_enc.7.weight"], P["vq_enc.7.bias"], padding=1)


def quantize_distances(P, x_vq):
    """x_vq [B,768,n] -> (x8 [B,n,8], d [B,n,bins]) with d = x^2 - 2 x.e + e^2 in fp32 (core_vq.py:175-183 up to the sign)."""
    x = ops.linear(np.ascontiguousarray(x_vq.transpose(0, 2, 1)), P["quantizer.vq.layers.0.project_in.weight"],
                   P["quantizer.vq.layers.0.project_in.bias"])
    e = P["quantizer.vq.layers.0._codebook.embed"].astype(F32)
    d = (np.square(x).sum(-1, keepdims=True, dtype=F32) - F32(2) * (x @ e.T) + np.square(e).sum(1, dtype=F32)[None, None, :]).astype(F32)
    return x, d


def encode(P, y):
    """SynthesizerTrn.encode (vqvae/model_24k.py:877-880): y [B,128,T] -> (codes [B, T/4] int64, x_vq [B,768,T/4])."""
    x_vq = vq_enc(P, np.asarray(y, F32))
    _, d = quantize_distances(P, x_vq)
    return d.argmin(-1).astype(np.int64), x_vq
