"""Weight packer: folded reference tensors -> kernel-ready blob for libdetail_hip.so.

Layouts (must match detail_tts_amd/csrc/conv_gemm.h and model.hip):
  * every Conv1d / Linear / HF Conv1D becomes a K-major GEMM operand
        wp[tap][ci][co]  (shape [KW, CinP, CoutP], CinP = ceil16(Cin), CoutP = packed_cout(rows), zero padded)
    and a bias `bp[CoutP]` in packed row order;
  * gated convs (WN in_layers: tanh/sigmoid halves, Conv1dGLU) interleave their output rows so that
    packed rows (2r, 2r+1) = (a_r, b_r) land in adjacent MFMA accumulator registers;
  * ConvTranspose1d(k, s, p) becomes a `phases = s` correlation: packed row = ph*Cout + co,
        y[co, q*s + ph] = sum_d sum_ci x[ci, q + d] * w[ci, co, ph + p - d*s],   d in [dmin, dmax]
    stored as taps d - dmin with `pad = -dmin`;
  * AttentionBlock relative-position bias (vqvae/utils/xtransformers.py:146-186) becomes a per-head table over
    clamp(s - t, -64, 64): bias_tab[h][off + 64] = emb[bucket(off)][h] * sqrt(head_dim).

These are weight-only transforms done once at load time on the host (like folding weight-norm).
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np

from .config import load_config

F32 = np.float32


def ceil_to(a, m):
    return (a + m - 1) // m * m


def packed_cout(rows):
    return ceil_to(rows, 128) if rows > 64 else (64 if rows > 32 else 32)


def pack_conv(w, b=None, row_perm=None):
    """w [Cout, Cin, k] (torch Conv1d layout) -> (wp [k, CinP, CoutP], bp [CoutP] | None)."""
    w = np.asarray(w, F32)
    if w.ndim == 2:
        w = w[:, :, None]
    if row_perm is not None:
        w = w[row_perm]
        if b is not None:
            b = np.asarray(b, F32)[row_perm]
    cout, cin, k = w.shape
    cinp, coutp = ceil_to(cin, 16), packed_cout(cout)
    wp = np.zeros((k, cinp, coutp), F32)
    wp[:, :cin, :cout] = w.transpose(2, 1, 0)
    bp = None
    if b is not None:
        bp = np.zeros((coutp,), F32)
        bp[:cout] = b
    return wp, bp


def gate_perm(n_rows):
    """[a_0..a_{h-1}, b_0..b_{h-1}] -> [a_0, b_0, a_1, b_1, ...]"""
    h = n_rows // 2
    perm = np.empty(n_rows, np.int64)
    perm[0::2] = np.arange(h)
    perm[1::2] = h + np.arange(h)
    return perm


def convtranspose_as_phases(w, stride, padding, output_padding=0):
    """w [Cin, Cout, k] -> (w_eq [stride*Cout, Cin, KW], pad) for the phase decomposition (output length T*stride)."""
    w = np.asarray(w, F32)
    cin, cout, k = w.shape
    if k - stride + output_padding != 2 * padding:
        raise ValueError("only ConvTranspose1d whose output length is T*stride (k - s + output_padding == 2p) is supported")
    dmin = -((k - 1 - padding) // stride)           # ceil((p-k+1)/s)
    dmax = (stride - 1 + padding) // stride
    kw = dmax - dmin + 1
    weq = np.zeros((stride * cout, cin, kw), F32)
    for ph in range(stride):
        for d in range(dmin, dmax + 1):
            j = ph + padding - d * stride
            if 0 <= j < k:
                weq[ph * cout:(ph + 1) * cout, :, d - dmin] = w[:, :, j].T
    return weq, -dmin


def rel_bucket(rel, num_buckets=32, max_distance=64):
    """RelativePositionBias._relative_position_bucket(causal=False); float32 log like torch."""
    rel = np.asarray(rel, np.int64)
    n = -rel
    nb = num_buckets // 2
    ret = (n < 0).astype(np.int64) * nb
    n = np.abs(n)
    max_exact = nb // 2
    is_small = n < max_exact
    with np.errstate(divide="ignore"):
        val = np.log(n.astype(F32) / F32(max_exact)) / F32(math.log(max_distance / max_exact)) * F32(nb - max_exact)
    val_large = np.minimum(max_exact + np.where(is_small, 0, val).astype(np.int64), nb - 1)
    return ret + np.where(is_small, n, val_large)


def bias_table(emb, head_dim, clip=64):
    """emb [32, H] -> [H, 2*clip+1] over offsets s - t in [-clip, clip] (beyond: same bucket)."""
    off = np.arange(-clip, clip + 1)
    b = rel_bucket(off)
    assert rel_bucket(np.array([clip]))[0] == rel_bucket(np.array([10 ** 6]))[0]
    assert rel_bucket(np.array([-clip]))[0] == rel_bucket(np.array([-10 ** 6]))[0]
    return np.ascontiguousarray((emb[b] * F32(math.sqrt(head_dim))).T, F32)


class Packer:
    def __init__(self):
        self.entries: "OrderedDict[str, np.ndarray]" = OrderedDict()

    def add(self, name, arr):
        assert name not in self.entries, name
        self.entries[name] = np.ascontiguousarray(arr, F32)

    def conv(self, name, w, b=None, row_perm=None, out_name=None):
        wp, bp = pack_conv(w, b, row_perm)
        out_name = out_name or name
        self.add(out_name + ".wp", wp)
        if bp is not None:
            self.add(out_name + ".bp", bp)

    def blob(self):
        """-> (flat fp32 array, names, offsets, numels); every tensor 64-byte aligned."""
        names, offsets, numels, total = [], [], [], 0
        for k, v in self.entries.items():
            names.append(k)
            offsets.append(total)
            numels.append(v.size)
            total += ceil_to(v.size, 16)
        flat = np.zeros(total, F32)
        for k, o, n in zip(names, offsets, numels):
            flat[o:o + n] = self.entries[k].reshape(-1)
        return flat, names, np.array(offsets, np.uint64), np.array(numels, np.uint64)


def _attention_block(pk, P, p, ch, heads):
    pk.add(p + ".norm.weight", P[p + ".norm.weight"])
    pk.add(p + ".norm.bias", P[p + ".norm.bias"])
    pk.conv(p + ".qkv", P[p + ".qkv.weight"], P[p + ".qkv.bias"])
    pk.conv(p + ".proj_out", P[p + ".proj_out.weight"], P[p + ".proj_out.bias"])
    pk.add(p + ".bias_tab", bias_table(P[p + ".relative_pos_embeddings.relative_attention_bias.weight"], ch // heads))


def _diff_resblock(pk, P, p):
    for n in ("in_layers.0.weight", "in_layers.0.bias", "out_layers.0.weight", "out_layers.0.bias"):
        pk.add(f"{p}.{n}", P[f"{p}.{n}"])
    pk.conv(p + ".in_layers.2", P[p + ".in_layers.2.weight"], P[p + ".in_layers.2.bias"])
    pk.conv(p + ".emb_layers.1", P[p + ".emb_layers.1.weight"], P[p + ".emb_layers.1.bias"])
    pk.conv(p + ".out_layers.3", P[p + ".out_layers.3.weight"], P[p + ".out_layers.3.bias"])


def pack_diffusion(pk, P, cfg):
    d = cfg["diffusion"]
    mc, heads, nl = d["model_channels"], d["num_heads"], d["num_layers"]
    pk.add("diffusion.unconditioned_embedding", P["diffusion.unconditioned_embedding"].reshape(-1))
    pk.conv("diffusion.inp_block", P["diffusion.inp_block.weight"], P["diffusion.inp_block.bias"])
    for i in (0, 2):
        pk.conv(f"diffusion.time_embed.{i}", P[f"diffusion.time_embed.{i}.weight"], P[f"diffusion.time_embed.{i}.bias"])
    pk.add("diffusion.code_norm.weight", P["diffusion.code_norm.weight"])
    pk.add("diffusion.code_norm.bias", P["diffusion.code_norm.bias"])
    pk.conv("diffusion.latent_conditioner.0", P["diffusion.latent_conditioner.0.weight"], P["diffusion.latent_conditioner.0.bias"])
    for i in range(1, 5):
        _attention_block(pk, P, f"diffusion.latent_conditioner.{i}", mc, heads)
    pk.conv("diffusion.contextual_embedder.0", P["diffusion.contextual_embedder.0.weight"], P["diffusion.contextual_embedder.0.bias"])
    pk.conv("diffusion.contextual_embedder.1", P["diffusion.contextual_embedder.1.weight"], P["diffusion.contextual_embedder.1.bias"])
    for i in range(2, 7):
        _attention_block(pk, P, f"diffusion.contextual_embedder.{i}", 2 * mc, heads)
    for i in range(3):
        p = f"diffusion.conditioning_timestep_integrator.{i}"
        _diff_resblock(pk, P, p + ".resblk")
        _attention_block(pk, P, p + ".attn", mc, heads)
    w = P["diffusion.integrating_conv.weight"]
    pk.conv("diffusion.integrating_conv.a", w[:, :mc], P["diffusion.integrating_conv.bias"])
    pk.conv("diffusion.integrating_conv.b", w[:, mc:], None)
    for i in range(nl):
        p = f"diffusion.layers.{i}"
        _diff_resblock(pk, P, p + ".resblk")
        _attention_block(pk, P, p + ".attn", mc, heads)
    for i in range(nl, nl + 3):
        _diff_resblock(pk, P, f"diffusion.layers.{i}")
    pk.add("diffusion.out.0.weight", P["diffusion.out.0.weight"])
    pk.add("diffusion.out.0.bias", P["diffusion.out.0.bias"])
    pk.conv("diffusion.out.2", P["diffusion.out.2.weight"], P["diffusion.out.2.bias"])


def _mel_style(pk, P, p):
    pk.conv(p + ".spectral.0.fc", P[p + ".spectral.0.fc.weight"], P[p + ".spectral.0.fc.bias"])
    pk.conv(p + ".spectral.3.fc", P[p + ".spectral.3.fc.weight"], P[p + ".spectral.3.fc.bias"])
    for i in range(2):
        w = P[p + f".temporal.{i}.conv1.conv.weight"]
        pk.conv(p + f".temporal.{i}.conv1.conv", w, P[p + f".temporal.{i}.conv1.conv.bias"], row_perm=gate_perm(w.shape[0]))
    wq = np.concatenate([P[p + f".slf_attn.{n}.weight"] for n in ("w_qs", "w_ks", "w_vs")], 0)
    bq = np.concatenate([P[p + f".slf_attn.{n}.bias"] for n in ("w_qs", "w_ks", "w_vs")], 0)
    pk.conv(p + ".slf_attn.qkv", wq, bq)
    pk.conv(p + ".slf_attn.fc", P[p + ".slf_attn.fc.weight"], P[p + ".slf_attn.fc.bias"])
    pk.conv(p + ".fc.fc", P[p + ".fc.fc.weight"], P[p + ".fc.fc.bias"])


def pack_vocoder(pk, P, cfg):
    v = cfg["vaegan"]
    hid = v["hidden_channels"]
    _mel_style(pk, P, "ref_enc")
    pk.conv("in_proj", P["in_proj.weight"], P["in_proj.bias"])
    for i in range(v["n_layers"]):
        a = f"enc_p.encoder.attn_layers.{i}"
        wq = np.concatenate([P[f"{a}.{n}.weight"] for n in ("conv_q", "conv_k", "conv_v")], 0)
        bq = np.concatenate([P[f"{a}.{n}.bias"] for n in ("conv_q", "conv_k", "conv_v")], 0)
        pk.conv(a + ".qkv", wq, bq)
        pk.conv(a + ".conv_o", P[a + ".conv_o.weight"], P[a + ".conv_o.bias"])
        pk.add(a + ".emb_rel_k", P[a + ".emb_rel_k"][0])
        pk.add(a + ".emb_rel_v", P[a + ".emb_rel_v"][0])
        for n in ("norm_layers_1", "norm_layers_2"):
            pk.add(f"enc_p.encoder.{n}.{i}.gamma", P[f"enc_p.encoder.{n}.{i}.gamma"])
            pk.add(f"enc_p.encoder.{n}.{i}.beta", P[f"enc_p.encoder.{n}.{i}.beta"])
        for n in ("conv_1", "conv_2"):
            pk.conv(f"enc_p.encoder.ffn_layers.{i}.{n}", P[f"enc_p.encoder.ffn_layers.{i}.{n}.weight"], P[f"enc_p.encoder.ffn_layers.{i}.{n}.bias"])
    pk.conv("enc_p.out_proj", P["enc_p.out_proj.weight"], P["enc_p.out_proj.bias"])
    pk.conv("enc_p.proj", P["enc_p.proj.weight"], P["enc_p.proj.bias"])
    for f in (0, 2, 4, 6):
        p = f"flow.flows.{f}"
        pk.conv(p + ".pre", P[p + ".pre.weight"], P[p + ".pre.bias"])
        pk.conv(p + ".post", P[p + ".post.weight"], P[p + ".post.bias"])
        gp = gate_perm(2 * hid)
        cperm = np.concatenate([l * 2 * hid + gp for l in range(4)])
        pk.conv(p + ".enc.cond_layer", P[p + ".enc.cond_layer.weight"], P[p + ".enc.cond_layer.bias"], row_perm=cperm)
        for l in range(4):
            pk.conv(p + f".enc.in_layers.{l}", P[p + f".enc.in_layers.{l}.weight"], P[p + f".enc.in_layers.{l}.bias"], row_perm=gp)
            w, b = P[p + f".enc.res_skip_layers.{l}.weight"], P[p + f".enc.res_skip_layers.{l}.bias"]
            if l < 3:
                pk.conv(p + f".enc.res_skip_layers.{l}.res", w[:hid], b[:hid])
                pk.conv(p + f".enc.res_skip_layers.{l}.skip", w[hid:], b[hid:])
            else:
                pk.conv(p + f".enc.res_skip_layers.{l}.skip", w, b)
    pk.conv("dec.conv_pre", P["dec.conv_pre.weight"], P["dec.conv_pre.bias"])
    pk.conv("dec.cond", P["dec.cond.weight"], P["dec.cond.bias"])
    for i, (u, k) in enumerate(zip(v["upsample_rates"], v["upsample_kernel_sizes"])):
        weq, _pad = convtranspose_as_phases(P[f"dec.ups.{i}.weight"], u, (k - u) // 2)
        pk.conv(f"dec.ups.{i}", weq, np.tile(P[f"dec.ups.{i}.bias"], u))
    nk = len(v["resblock_kernel_sizes"])
    for i in range(len(v["upsample_rates"]) * nk):
        for cs in ("convs1", "convs2"):
            for l in range(3):
                p = f"dec.resblocks.{i}.{cs}.{l}"
                pk.conv(p, P[p + ".weight"], P[p + ".bias"])
    pk.conv("dec.conv_post", P["dec.conv_post.weight"], None)


def pack_gpt(pk, P, cfg):
    g = cfg["gpt"]
    _mel_style(pk, P, "gpt.conditioning_encoder")
    for l in range(g["layers"]):
        p = f"gpt.gpt.h.{l}"
        for n in ("ln_1", "ln_2"):
            pk.add(f"{p}.{n}.weight", P[f"{p}.{n}.weight"])
            pk.add(f"{p}.{n}.bias", P[f"{p}.{n}.bias"])
        for n in ("attn.c_attn", "attn.c_proj", "mlp.c_fc", "mlp.c_proj"):      # HF Conv1D: weight [in, out]
            pk.conv(f"{p}.{n}", P[f"{p}.{n}.weight"].T, P[f"{p}.{n}.bias"])
    for n in ("gpt.gpt.ln_f", "gpt.final_norm"):
        pk.add(n + ".weight", P[n + ".weight"])
        pk.add(n + ".bias", P[n + ".bias"])
    pk.conv("gpt.mel_head", P["gpt.mel_head.weight"], P["gpt.mel_head.bias"])
    for n in ("gpt.text_embedding.weight", "gpt.mel_embedding.weight", "gpt.text_pos_embedding.emb.weight",
              "gpt.mel_pos_embedding.emb.weight"):
        pk.add(n, P[n])


def pack_vq(pk, P, cfg):
    """infer_gpt's decode path: quantizer.decode -> + vq_ref_enc -> vq_dec  (vqvae/model_24k.py:828-845)."""
    emb = P["quantizer.vq.layers.0._codebook.embed"].astype(np.float64)
    wo = P["quantizer.vq.layers.0.project_out.weight"].astype(np.float64)
    # EuclideanCodebook.dequantize + project_out folded into one [bins, 768] lookup table (core_vq.py:188-190, 298-301)
    pk.add("quantizer.table", (emb @ wo.T + P["quantizer.vq.layers.0.project_out.bias"]).astype(F32))
    pk.add("vq_dec.1.weight", P["vq_dec.1.weight"])
    pk.add("vq_dec.1.bias", P["vq_dec.1.bias"])
    for i in (3, 5):
        weq, _pad = convtranspose_as_phases(P[f"vq_dec.{i}.weight"], 2, 1, output_padding=1)
        pk.conv(f"vq_dec.{i}", weq, np.tile(P[f"vq_dec.{i}.bias"], 2))
    pk.conv("vq_dec.7", P["vq_dec.7.weight"], P["vq_dec.7.bias"])
    _mel_style(pk, P, "vq_ref_enc")
    # encode side (vqvae/model_24k.py:877-880): vq_enc, project_in, and the codebook with its squared norms for the nearest search
    if "vq_enc.3.weight" in P:
        pk.add("vq_enc.1.weight", P["vq_enc.1.weight"])
        pk.add("vq_enc.1.bias", P["vq_enc.1.bias"])
        for i in (3, 5, 7):
            pk.conv(f"vq_enc.{i}", P[f"vq_enc.{i}.weight"], P[f"vq_enc.{i}.bias"])
        pk.conv("quantizer.project_in", P["quantizer.vq.layers.0.project_in.weight"][:, :, None], P["quantizer.vq.layers.0.project_in.bias"])
        e = P["quantizer.vq.layers.0._codebook.embed"].astype(F32)
        pk.add("quantizer.embed", e)
        pk.add("quantizer.embed_sq", np.square(e).sum(1, dtype=F32))          # embed.pow(2).sum(0) of core_vq.py:180, fp32 like torch


def pack_all(P, cfg=None, parts=("diffusion",)):
    """P: folded fp32 dict (weights.select_inference_params). Returns a Packer."""
    cfg = load_config(cfg)
    pk = Packer()
    if "diffusion" in parts:
        pack_diffusion(pk, P, cfg)
    if "vocoder" in parts:
        pack_vocoder(pk, P, cfg)
    if "gpt" in parts:
        pack_gpt(pk, P, cfg)
    if "vq" in parts:
        pack_vq(pk, P, cfg)
    if "frontend" in parts:
        from .frontend import pack_frontend
        pack_frontend(pk, cfg)
    return pk
