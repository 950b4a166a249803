"""Oracle for stage B: the diffusion mel decoder (TEST INFRASTRUCTURE).

Restates vqvae/diff_model.py, vqvae/utils/diff_util.py:113-215,
vqvae/utils/xtransformers.py:146-186, vqvae/utils/diffusion.py (sampler subset)
and vqvae/model_24k.py:479-509 in numpy.  `P` is a dict of *folded* fp32 weights
keyed by the reference's state-dict names (detail_tts_amd/weights.py).
"""
from __future__ import annotations

import math

import numpy as np

from . import ops, philox

F32 = np.float32


# ----------------------------------------------------------------------------
# schedule (vqvae/utils/diffusion.py:83-98, 179-228, 1181-1195, 1223-1272)
# ----------------------------------------------------------------------------
def space_timesteps(num_timesteps, section_count):
    """space_timesteps(num, [section_count]) with a single section."""
    frac_stride = 1 if section_count <= 1 else (num_timesteps - 1) / (section_count - 1)
    cur, taken = 0.0, []
    for _ in range(section_count):
        taken.append(round(cur))
        cur += frac_stride
    return sorted(set(taken))


def make_schedule(trained_steps=4000, steps=50):
    """All float64 tables of SpacedDiffusion(space_timesteps(4000,[50]), linear betas)."""
    scale = 1000 / trained_steps
    betas = np.linspace(scale * 0.0001, scale * 0.02, trained_steps, dtype=np.float64)
    ac = np.cumprod(1.0 - betas)
    use = space_timesteps(trained_steps, steps)
    last, new_betas, tmap = 1.0, [], []
    for i, a in enumerate(ac):
        if i in set(use):
            new_betas.append(1 - a / last)
            last = a
            tmap.append(i)
    b = np.array(new_betas, dtype=np.float64)
    alphas = 1.0 - b
    acp = np.cumprod(alphas)
    acp_prev = np.append(1.0, acp[:-1])
    post_var = b * (1.0 - acp_prev) / (1.0 - acp)
    return {
        "timestep_map": np.array(tmap, np.int64),
        "betas": b,
        "log_betas": np.log(b),
        "sqrt_recip_alphas_cumprod": np.sqrt(1.0 / acp),
        "sqrt_recipm1_alphas_cumprod": np.sqrt(1.0 / acp - 1),
        "posterior_log_variance_clipped": np.log(np.append(post_var[1], post_var[1:])),
        "posterior_mean_coef1": b * np.sqrt(acp_prev) / (1.0 - acp),
        "posterior_mean_coef2": (1.0 - acp_prev) * np.sqrt(alphas) / (1.0 - acp),
        "num_timesteps": len(b),
    }


# ----------------------------------------------------------------------------
# relative position bias (vqvae/utils/xtransformers.py:146-186)
# ----------------------------------------------------------------------------
def rel_bucket(rel, num_buckets=32, max_distance=64):
    """bucket of rel = k_pos - q_pos (non-causal). float32 log like torch."""
    rel = np.asarray(rel, np.int64)
    n = -rel
    nb = num_buckets // 2
    ret = (n < 0).astype(np.int64) * nb
    n = np.abs(n)
    max_exact = nb // 2
    is_small = n < max_exact
    with np.errstate(divide="ignore"):
        val = np.log(n.astype(F32) / F32(max_exact)) / F32(math.log(max_distance / max_exact)) * F32(nb - max_exact)
    val_large = max_exact + np.where(is_small, 0, val).astype(np.int64)
    val_large = np.minimum(val_large, nb - 1)
    return ret + np.where(is_small, n, val_large)


_REL_BIAS_CACHE = {}


def rel_bias(table, T, scale):
    """[H,T,T] additive bias: table[bucket(j-i), h] * scale. table [32,H].  Memoised per (table, T): the weights are fixed and the
    reference recomputes the same tensor in every forward."""
    key = (table.ctypes.data, table.shape, T, float(scale))
    if key in _REL_BIAS_CACHE:
        return _REL_BIAS_CACHE[key]
    out = _REL_BIAS_CACHE[key] = _rel_bias(table, T, scale)
    if len(_REL_BIAS_CACHE) > 64:
        _REL_BIAS_CACHE.pop(next(iter(_REL_BIAS_CACHE)))
    return out


def _rel_bias(table, T, scale):
    pos = np.arange(T)
    bucket = rel_bucket(pos[None, :] - pos[:, None])
    return (table[bucket].transpose(2, 0, 1) * F32(scale)).astype(F32)


# ----------------------------------------------------------------------------
# blocks
# ----------------------------------------------------------------------------
def attention_block(P, p, x, heads):
    """AttentionBlock.forward, vqvae/utils/diff_util.py:209-215 + QKVAttentionLegacy :146-169."""
    B, C, T = x.shape
    h = ops.group_norm(x, ops.gn_groups(C), P[p + ".norm.weight"], P[p + ".norm.bias"])
    qkv = ops.conv1d(h, P[p + ".qkv.weight"], P[p + ".qkv.bias"])
    ch = C // heads
    qkv = qkv.reshape(B * heads, 3 * ch, T)
    q, k, v = qkv[:, :ch], qkv[:, ch:2 * ch], qkv[:, 2 * ch:]
    scale = F32(1.0 / math.sqrt(math.sqrt(ch)))
    bias = rel_bias(P[p + ".relative_pos_embeddings.relative_attention_bias.weight"], T, ch ** 0.5)
    # "bct,bcs->bts", + bias, softmax, "bts,bcs->bct"
    a = ops.attention_weights_apply(q * scale, k * scale, v, bias, heads).reshape(B, C, T)
    return x + ops.conv1d(a, P[p + ".proj_out.weight"], P[p + ".proj_out.bias"])


def res_block(P, p, x, t_emb):
    """ResBlock.forward (use_scale_shift_norm, efficient_config), vqvae/diff_model.py:106-119."""
    C = x.shape[1]
    g = ops.gn_groups(C)
    h = ops.silu(ops.group_norm(x, g, P[p + ".in_layers.0.weight"], P[p + ".in_layers.0.bias"]))
    h = ops.conv1d(h, P[p + ".in_layers.2.weight"], P[p + ".in_layers.2.bias"])
    emb = ops.linear(ops.silu(t_emb), P[p + ".emb_layers.1.weight"], P[p + ".emb_layers.1.bias"])
    scale, shift = emb[:, :C, None], emb[:, C:, None]
    h = ops.group_norm(h, g, P[p + ".out_layers.0.weight"], P[p + ".out_layers.0.bias"]) * (1 + scale) + shift
    h = ops.conv1d(ops.silu(h.astype(F32)), P[p + ".out_layers.3.weight"], P[p + ".out_layers.3.bias"], padding=1)
    return x + h


def diffusion_layer(P, p, x, t_emb, heads):
    """DiffusionLayer.forward, vqvae/diff_model.py:128-130."""
    return attention_block(P, p + ".attn", res_block(P, p + ".resblk", x, t_emb), heads)


def timestep_embedding(ts, dim, max_period=10000):
    """vqvae/diff_model.py:20-38 (cos half first)."""
    half = dim // 2
    freqs = np.exp(-math.log(max_period) * np.arange(half, dtype=F32) / F32(half)).astype(F32)
    args = np.asarray(ts, F32)[:, None] * freqs[None]
    return np.concatenate([np.cos(args), np.sin(args)], -1).astype(F32)


def time_embed(P, ts, mc):
    e = timestep_embedding(ts, mc)
    e = ops.silu(ops.linear(e, P["diffusion.time_embed.0.weight"], P["diffusion.time_embed.0.bias"]))
    return ops.linear(e, P["diffusion.time_embed.2.weight"], P["diffusion.time_embed.2.bias"])


def get_conditioning(P, mel, heads=16):
    """DiffusionTts.get_conditioning, vqvae/diff_model.py:221-229 (single clip)."""
    p = "diffusion.contextual_embedder"
    h = ops.conv1d(mel, P[p + ".0.weight"], P[p + ".0.bias"], stride=2, padding=1)
    h = ops.conv1d(h, P[p + ".1.weight"], P[p + ".1.bias"], stride=2, padding=1)
    for i in range(2, 7):
        h = attention_block(P, f"{p}.{i}", h, heads)
    return h.mean(-1).astype(F32)


def timestep_independent(P, latent, cond_latent, seq_len, heads=16):
    """DiffusionTts.timestep_independent (latent branch, eval), vqvae/diff_model.py:231-260.
    latent [B,n,768] -> [B,768,seq_len]."""
    x = np.ascontiguousarray(latent.transpose(0, 2, 1))
    C = x.shape[1]
    p = "diffusion.latent_conditioner"
    h = ops.conv1d(x, P[p + ".0.weight"], P[p + ".0.bias"], padding=1)
    for i in range(1, 5):
        h = attention_block(P, f"{p}.{i}", h, heads)
    scale, shift = cond_latent[:, :C, None], cond_latent[:, C:, None]
    h = ops.group_norm(h, ops.gn_groups(C), P["diffusion.code_norm.weight"], P["diffusion.code_norm.bias"]) * (1 + scale) + shift
    # F.interpolate(mode='nearest', size=seq_len): src = floor(dst * in/out)
    n = h.shape[-1]
    idx = np.floor(np.arange(seq_len) * (n / seq_len)).astype(np.int64)
    return np.ascontiguousarray(h[:, :, idx], F32)


def diffusion_forward(P, x, ts, code_emb=None, conditioning_free=False, heads=16, num_layers=10):
    """DiffusionTts.forward with precomputed_aligned_embeddings, vqvae/diff_model.py:262-322.
    x [B,128,T]; ts [B] (already mapped to the 4000-step scale) -> [B,256,T]."""
    B, _, T = x.shape
    mc = P["diffusion.inp_block.weight"].shape[0]
    if conditioning_free:
        code_emb = np.broadcast_to(P["diffusion.unconditioned_embedding"], (B, mc, T)).astype(F32)
    t_emb = time_embed(P, ts, mc)
    c = code_emb
    for i in range(3):
        c = diffusion_layer(P, f"diffusion.conditioning_timestep_integrator.{i}", c, t_emb, heads)
    h = ops.conv1d(x, P["diffusion.inp_block.weight"], P["diffusion.inp_block.bias"], padding=1)
    h = np.concatenate([h, c], 1)
    h = ops.conv1d(h, P["diffusion.integrating_conv.weight"], P["diffusion.integrating_conv.bias"])
    for i in range(num_layers):
        h = diffusion_layer(P, f"diffusion.layers.{i}", h, t_emb, heads)
    for i in range(num_layers, num_layers + 3):
        h = res_block(P, f"diffusion.layers.{i}", h, t_emb)
    h = ops.silu(ops.group_norm(h, ops.gn_groups(mc), P["diffusion.out.0.weight"], P["diffusion.out.0.bias"]))
    return ops.conv1d(h, P["diffusion.out.2.weight"], P["diffusion.out.2.bias"], padding=1)


# ----------------------------------------------------------------------------
# sampler (vqvae/utils/diffusion.py:284-386, 445-485, 654-742)
# ----------------------------------------------------------------------------
def p_sample_update(sched, i, x, out_c, out_u, noise, cond_free_k=2.0):
    """One ancestral step given the two model outputs. Returns (x_next, pred_xstart)."""
    C = x.shape[1]
    eps_c, var_v = out_c[:, :C], out_c[:, C:]
    eps_u = out_u[:, :C]
    min_log = F32(sched["posterior_log_variance_clipped"][i])
    max_log = F32(sched["log_betas"][i])
    frac = (var_v + 1) / 2
    log_var = frac * max_log + (1 - frac) * min_log
    cfk = cond_free_k * (1 - i / sched["num_timesteps"])
    eps = (1 + cfk) * eps_c - cfk * eps_u
    x0 = F32(sched["sqrt_recip_alphas_cumprod"][i]) * x - F32(sched["sqrt_recipm1_alphas_cumprod"][i]) * eps
    x0 = np.clip(x0, -1, 1)
    mean = F32(sched["posterior_mean_coef1"][i]) * x0 + F32(sched["posterior_mean_coef2"][i]) * x
    nz = 0.0 if i == 0 else 1.0
    return (mean + nz * np.exp(0.5 * log_var) * noise).astype(F32), x0.astype(F32)


def p_sample_loop(P, sched, code_emb, noise0, step_noise, cond_free_k=2.0, n_steps=None, trace=None):
    """p_sample_loop with classifier-free guidance.  noise0 [B,128,T]; step_noise(i)->[B,128,T].
    n_steps limits the number of steps executed (tests)."""
    x = noise0.astype(F32)
    B = x.shape[0]
    idx = list(range(sched["num_timesteps"]))[::-1]
    if n_steps is not None:
        idx = idx[:n_steps]
    for i in idx:
        ts = np.full((B,), sched["timestep_map"][i], np.int64)
        out_c = diffusion_forward(P, x, ts, code_emb)
        out_u = diffusion_forward(P, x, ts, conditioning_free=True)
        x_new, _ = p_sample_update(sched, i, x, out_c, out_u, step_noise(i), cond_free_k)
        if trace is not None:
            trace.append({"i": i, "x_in": x, "out_c": out_c, "out_u": out_u, "x_out": x_new})
        x = x_new
    return x


MEL_MIN = -11.512925465
TORCH_MEL_MAX = 2.7


def denormalize_mel(m):
    """vqvae/model_24k.py:508-509."""
    return (((m + 1) / 2) * (TORCH_MEL_MAX - MEL_MIN) + MEL_MIN).astype(F32)


def do_spectrogram_diffusion(P, sched, latent, cond_latent, seed, sample_ids, temperature=1.0, n_steps=None):
    """vqvae/model_24k.py:479-492 with the Philox noise spec (oracle/philox.py)."""
    B, n, _ = latent.shape
    T = 4 * n
    code_emb = timestep_independent(P, latent, cond_latent, T)
    noise0 = np.stack([philox.normal(seed, s, philox.STAGE_DIFF_INIT, 0, 128 * T).reshape(128, T) for s in sample_ids]) * F32(temperature)

    def step_noise(i):
        return np.stack([philox.normal(seed, s, philox.STAGE_DIFF_STEP, i, 128 * T).reshape(128, T) for s in sample_ids])

    return p_sample_loop(P, sched, code_emb, noise0, step_noise, n_steps=n_steps)
