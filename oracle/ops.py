"""numpy fp32 building blocks used by the oracle (TEST INFRASTRUCTURE).

Layouts follow the reference (PyTorch): activations [B, C, T]; Conv1d weight
[Cout, Cin, k]; ConvTranspose1d weight [Cin, Cout, k]; Linear weight [out, in].
"""
from __future__ import annotations

import numpy as np

F32 = np.float32

# ---- optional torch-CPU backend (bench.py's cpu_baseline leg).  The numpy forms below are the checker: fp64 statistics, single-threaded
# elementwise passes - faithful, slow (a DiffusionTts.forward at T = 936 spends 4 of its 4.6 s in softmax / astype copies).  With
# use_torch(True) the heavy building blocks run through torch's multi-threaded fp32 CPU kernels (oneDNN conv, fused softmax /
# normalisation) - the same primitives the reference itself runs on a CPU - while every oracle function keeps its own code and its
# numpy-in / numpy-out interface.  tests/test_oracle_golden.py checks this backend against the reference's fixtures too.
_TORCH = {"on": False}
_TORCH_MIN = 1 << 16          # below this many elements torch's thread-pool dispatch costs more than the numpy pass


def use_torch(on=True):
    prev = _TORCH["on"]
    _TORCH["on"] = bool(on)
    return prev


def _t(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a, F32))


def conv1d(x, w, b=None, stride=1, padding=0, dilation=1):
    """torch.nn.functional.conv1d semantics. x [B,Cin,T], w [Cout,Cin,k]."""
    if _TORCH["on"]:
        import torch
        return torch.nn.functional.conv1d(_t(x), _t(w), None if b is None else _t(b), stride, padding, dilation).numpy()
    x = np.asarray(x, F32)
    B, Cin, T = x.shape
    Cout, Cin2, k = w.shape
    assert Cin == Cin2
    if padding:
        x = np.pad(x, ((0, 0), (0, 0), (padding, padding)))
    Tp = x.shape[2]
    Tout = (Tp - dilation * (k - 1) - 1) // stride + 1
    # cols[b, tap, ci, t] = x[b, ci, t*stride + tap*dilation]
    s0, s1, s2 = x.strides
    cols = np.lib.stride_tricks.as_strided(
        x, shape=(B, k, Cin, Tout), strides=(s0, s2 * dilation, s1, s2 * stride), writeable=False)
    w2 = np.ascontiguousarray(w.transpose(0, 2, 1)).reshape(Cout, k * Cin)  # [Cout, tap*Cin+ci]
    out = np.empty((B, Cout, Tout), F32)
    for bi in range(B):
        out[bi] = w2 @ np.ascontiguousarray(cols[bi]).reshape(k * Cin, Tout)
    if b is not None:
        out += np.asarray(b, F32)[None, :, None]
    return out


def conv_transpose1d(x, w, b=None, stride=1, padding=0):
    """torch ConvTranspose1d (no output_padding, dilation 1). w [Cin,Cout,k].
    out[b,co,t*stride + j - padding] += x[b,ci,t] * w[ci,co,j]"""
    x = np.asarray(x, F32)
    B, Cin, T = x.shape
    _, Cout, k = w.shape
    Lfull = (T - 1) * stride + k
    full = np.zeros((B, Cout, Lfull), F32)
    for j in range(k):
        # contribution of tap j: [B,Cout,T] placed at positions t*stride + j
        contrib = np.einsum("io,bit->bot", w[:, :, j], x, optimize=True).astype(F32)
        full[:, :, j: j + (T - 1) * stride + 1: stride] += contrib
    out = full[:, :, padding: Lfull - padding] if padding else full
    if b is not None:
        out = out + np.asarray(b, F32)[None, :, None]
    return np.ascontiguousarray(out, F32)


def linear(x, w, b=None):
    """x [..., in], w [out, in]."""
    if _TORCH["on"]:
        import torch
        return torch.nn.functional.linear(_t(x), _t(w), None if b is None else _t(b)).numpy()
    y = np.asarray(x, F32) @ np.asarray(w, F32).T
    if b is not None:
        y = y + b
    return y.astype(F32)


def group_norm(x, groups, gamma, beta, eps=1e-5):
    """torch.nn.GroupNorm on [B,C,T] (biased variance); statistics in float64, apply in float32."""
    if _TORCH["on"]:
        import torch
        return torch.nn.functional.group_norm(_t(x), groups, _t(gamma), _t(beta), eps).numpy()
    B, C, T = x.shape
    xg = x.reshape(B, groups, -1)
    mean = xg.mean(-1, keepdims=True, dtype=np.float64)
    var = np.square(xg - mean.astype(F32)).mean(-1, keepdims=True, dtype=np.float64)
    rstd = (1.0 / np.sqrt(var + eps)).astype(F32)
    y = ((xg - mean.astype(F32)) * rstd).reshape(B, C, T)
    return (y * gamma[None, :, None] + beta[None, :, None]).astype(F32)


def gn_groups(channels):
    """`normalization()` group count, vqvae/utils/diff_util.py:118-133."""
    groups = 32
    if channels <= 16:
        groups = 8
    elif channels <= 64:
        groups = 16
    while channels % groups != 0:
        groups = int(groups / 2)
    assert groups > 2
    return groups


def layer_norm_last(x, gamma, beta, eps=1e-5):
    if _TORCH["on"] and np.size(x) >= _TORCH_MIN:
        import torch
        return torch.nn.functional.layer_norm(_t(x), (x.shape[-1],), _t(gamma), _t(beta), eps).numpy()
    x64 = x.astype(np.float64)
    mean = x64.mean(-1, keepdims=True)
    var = x64.var(-1, keepdims=True)
    return (((x64 - mean) / np.sqrt(var + eps)) * gamma + beta).astype(F32)


def layer_norm_channels(x, gamma, beta, eps=1e-5):
    """modules.LayerNorm on [B,C,T] (vqvae/modules/modules.py:36-48)."""
    return layer_norm_last(x.transpose(0, 2, 1), gamma, beta, eps).transpose(0, 2, 1)


def sigmoid(x):
    if _TORCH["on"] and np.size(x) >= _TORCH_MIN:
        import torch
        return torch.sigmoid(_t(x)).numpy()
    return (1.0 / (1.0 + np.exp(-np.asarray(x, F32)))).astype(F32)


def silu(x):
    if _TORCH["on"] and np.size(x) >= _TORCH_MIN:
        import torch
        return torch.nn.functional.silu(_t(x)).numpy()
    return (x * sigmoid(x)).astype(F32)


def mish(x):
    """x * tanh(softplus(x)), vqvae/modules/modules.py:497-502."""
    if _TORCH["on"] and np.size(x) >= _TORCH_MIN:
        import torch
        return torch.nn.functional.mish(_t(x)).numpy()
    x64 = x.astype(np.float64)
    sp = np.logaddexp(0.0, x64)
    return (x64 * np.tanh(sp)).astype(F32)


def gelu_new(x):
    if _TORCH["on"] and np.size(x) >= _TORCH_MIN:
        import torch
        return torch.nn.functional.gelu(_t(x), approximate="tanh").numpy()
    x64 = x.astype(np.float64)
    return (0.5 * x64 * (1.0 + np.tanh(np.sqrt(2.0 / np.pi) * (x64 + 0.044715 * x64 ** 3)))).astype(F32)


def leaky_relu(x, slope):
    if _TORCH["on"] and np.size(x) >= _TORCH_MIN:
        import torch
        return torch.nn.functional.leaky_relu(_t(x), float(slope)).numpy()
    return np.where(x >= 0, x, x * F32(slope)).astype(F32)


def softmax(x, axis=-1):
    if _TORCH["on"] and np.size(x) >= _TORCH_MIN:
        import torch
        return torch.softmax(_t(x), axis).numpy()
    x = np.asarray(x, F32)
    m = x.max(axis=axis, keepdims=True)
    m = np.where(np.isfinite(m), m, F32(0))
    e = np.exp(x - m)
    return (e / e.sum(axis=axis, keepdims=True, dtype=np.float64).astype(F32)).astype(F32)


_BIAS_T = {}


def attention_weights_apply(q, k, v, bias=None, heads=1):
    """softmax(q^T k + bias, -1) applied to v: q, k, v [N,c,T|S] (N = batch x heads), bias [heads,T,S] shared by the batch -> [N,c,T]
    (the 'bct,bcs->bts' / softmax / 'bts,bcs->bct' core of QKVAttentionLegacy, vqvae/utils/diff_util.py:146-169)."""
    N, c, T = q.shape
    S = k.shape[2]
    if _TORCH["on"]:
        import torch
        w = torch.matmul(_t(q).transpose(1, 2), _t(k))
        if bias is not None:
            key = (bias.ctypes.data, bias.shape)
            bt = _BIAS_T.get(key)
            if bt is None:
                if len(_BIAS_T) > 64:
                    _BIAS_T.clear()
                bt = _BIAS_T[key] = _t(bias)
            w = (w.view(N // heads, heads, T, S) + bt[None]).view(N, T, S)
        w = torch.softmax(w, -1)
        return torch.matmul(_t(v), w.transpose(1, 2)).numpy()
    w = np.matmul(q.transpose(0, 2, 1), k).astype(F32)
    if bias is not None:
        w = (w.reshape(N // heads, heads, T, S) + bias[None]).reshape(N, T, S)
    w = softmax(w, -1)
    return np.matmul(v, w.transpose(0, 2, 1)).astype(F32)


def sequence_mask(lengths, max_len):
    """vqvae/modules/commons.py:144-148."""
    return (np.arange(max_len)[None, :] < np.asarray(lengths)[:, None])
