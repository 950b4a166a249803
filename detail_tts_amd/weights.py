"""Inference weight inventory, synthetic weights, weight-norm folding.

The names are the reference's ``SynthesizerTrn.state_dict()`` keys restricted to
what ``SynthesizerTrn.infer`` touches (reference
constructors: vqvae/model_24k.py:515-650, gpt/model.py:265-331,
vqvae/diff_model.py:133-209, vqvae/modules/modules.py:152-229/240-313/421-455/
642-695, vqvae/modules/attentions.py:73-94/161-195/317-336).  A real checkpoint
(`ckpt['G']`, prepare/load_infer.py:21-26) is accepted as-is: unneeded keys
(`enc_q.*`, `quantizer.*`, `vq_*`, `gpt.text_head.*`, the `gpt.inference_model.*`
aliases ...) are ignored, and old-style weight-norm pairs
(`weight_g`/`weight_v`) are folded to a plain `weight`.

Nothing here touches the GPU; it is host-side numpy.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict

import numpy as np

from .config import load_config

# kinds drive the synthetic initialiser only
K_W = "w"          # conv / linear weight, fan-in scaled uniform
K_B = "b"          # bias of the layer named by the same prefix
K_NG = "norm_g"    # norm scale
K_NB = "norm_b"    # norm shift
K_EMB = "emb"      # embedding table N(0, .02)
K_HFW = "hf_w"     # HF GPT-2 Conv1D weight [in, out], N(0, .02)
K_HFB = "hf_b"
K_WN_V = "wn_v"    # weight-norm direction
K_WN_G = "wn_g"    # weight-norm magnitude
K_RELB = "relbias"
K_RELE = "relemb"
K_UNCOND = "uncond"


def _attention_block(spec, p, ch, heads):
    spec[p + ".norm.weight"] = ((ch,), K_NG)
    spec[p + ".norm.bias"] = ((ch,), K_NB)
    spec[p + ".qkv.weight"] = ((3 * ch, ch, 1), K_W)
    spec[p + ".qkv.bias"] = ((3 * ch,), K_B)
    spec[p + ".proj_out.weight"] = ((ch, ch, 1), K_W)
    spec[p + ".proj_out.bias"] = ((ch,), K_B)
    spec[p + ".relative_pos_embeddings.relative_attention_bias.weight"] = ((32, heads), K_RELB)


def _diff_resblock(spec, p, ch):
    spec[p + ".in_layers.0.weight"] = ((ch,), K_NG)
    spec[p + ".in_layers.0.bias"] = ((ch,), K_NB)
    spec[p + ".in_layers.2.weight"] = ((ch, ch, 1), K_W)
    spec[p + ".in_layers.2.bias"] = ((ch,), K_B)
    spec[p + ".emb_layers.1.weight"] = ((2 * ch, ch), K_W)
    spec[p + ".emb_layers.1.bias"] = ((2 * ch,), K_B)
    spec[p + ".out_layers.0.weight"] = ((ch,), K_NG)
    spec[p + ".out_layers.0.bias"] = ((ch,), K_NB)
    spec[p + ".out_layers.3.weight"] = ((ch, ch, 3), K_W)
    spec[p + ".out_layers.3.bias"] = ((ch,), K_B)


def _diffusion_layer(spec, p, ch, heads):
    _diff_resblock(spec, p + ".resblk", ch)
    _attention_block(spec, p + ".attn", ch, heads)


def _mel_style_encoder(spec, p, n_mel, hidden, out):
    spec[p + ".spectral.0.fc.weight"] = ((hidden, n_mel), K_W)
    spec[p + ".spectral.0.fc.bias"] = ((hidden,), K_B)
    spec[p + ".spectral.3.fc.weight"] = ((hidden, hidden), K_W)
    spec[p + ".spectral.3.fc.bias"] = ((hidden,), K_B)
    for i in range(2):
        spec[p + f".temporal.{i}.conv1.conv.weight"] = ((2 * hidden, hidden, 5), K_W)
        spec[p + f".temporal.{i}.conv1.conv.bias"] = ((2 * hidden,), K_B)
    for n in ("w_qs", "w_ks", "w_vs", "fc"):
        spec[p + f".slf_attn.{n}.weight"] = ((hidden, hidden), K_W)
        spec[p + f".slf_attn.{n}.bias"] = ((hidden,), K_B)
    spec[p + ".fc.fc.weight"] = ((out, hidden), K_W)
    spec[p + ".fc.fc.bias"] = ((out,), K_B)


def inference_param_spec(cfg=None) -> "OrderedDict[str, tuple]":
    """name -> (shape, kind) for every tensor `infer` needs, in state-dict form
    (i.e. weight-norm layers appear as weight_g / weight_v pairs)."""
    cfg = load_config(cfg)
    spec: "OrderedDict[str, tuple]" = OrderedDict()
    d, g, v = cfg["diffusion"], cfg["gpt"], cfg["vaegan"]
    n_mel = cfg["data"]["n_mel_channels"]

    # ---- dec (Generator, vqvae/model_24k.py:221-267)
    c0 = v["upsample_initial_channel"]
    inter = v["inter_channels"]
    gin = v["gin_channels"]
    spec["dec.conv_pre.weight"] = ((c0, inter, 7), K_W)
    spec["dec.conv_pre.bias"] = ((c0,), K_B)
    for i, (u, k) in enumerate(zip(v["upsample_rates"], v["upsample_kernel_sizes"])):
        cin, cout = c0 // (2 ** i), c0 // (2 ** (i + 1))
        spec[f"dec.ups.{i}.bias"] = ((cout,), K_B)
        spec[f"dec.ups.{i}.weight_g"] = ((cin, 1, 1), K_WN_G)
        spec[f"dec.ups.{i}.weight_v"] = ((cin, cout, k), K_WN_V)
    nk = len(v["resblock_kernel_sizes"])
    for i in range(len(v["upsample_rates"])):
        ch = c0 // (2 ** (i + 1))
        for j, k in enumerate(v["resblock_kernel_sizes"]):
            for cs in ("convs1", "convs2"):
                for l in range(3):
                    p = f"dec.resblocks.{i * nk + j}.{cs}.{l}"
                    spec[p + ".bias"] = ((ch,), K_B)
                    spec[p + ".weight_g"] = ((ch, 1, 1), K_WN_G)
                    spec[p + ".weight_v"] = ((ch, ch, k), K_WN_V)
    spec["dec.conv_post.weight"] = ((1, ch, 7), K_W)
    spec["dec.cond.weight"] = ((c0, gin, 1), K_W)
    spec["dec.cond.bias"] = ((c0,), K_B)

    # ---- diffusion (DiffusionTts, vqvae/diff_model.py:133-209)
    mc, heads = d["model_channels"], d["num_heads"]
    spec["diffusion.unconditioned_embedding"] = ((1, mc, 1), K_UNCOND)
    spec["diffusion.inp_block.weight"] = ((mc, d["in_channels"], 3), K_W)
    spec["diffusion.inp_block.bias"] = ((mc,), K_B)
    for i in (0, 2):
        spec[f"diffusion.time_embed.{i}.weight"] = ((mc, mc), K_W)
        spec[f"diffusion.time_embed.{i}.bias"] = ((mc,), K_B)
    spec["diffusion.code_norm.weight"] = ((mc,), K_NG)
    spec["diffusion.code_norm.bias"] = ((mc,), K_NB)
    spec["diffusion.latent_conditioner.0.weight"] = ((mc, d["in_latent_channels"], 3), K_W)
    spec["diffusion.latent_conditioner.0.bias"] = ((mc,), K_B)
    for i in range(1, 5):
        _attention_block(spec, f"diffusion.latent_conditioner.{i}", mc, heads)
    spec["diffusion.contextual_embedder.0.weight"] = ((mc, d["in_channels"], 3), K_W)
    spec["diffusion.contextual_embedder.0.bias"] = ((mc,), K_B)
    spec["diffusion.contextual_embedder.1.weight"] = ((2 * mc, mc, 3), K_W)
    spec["diffusion.contextual_embedder.1.bias"] = ((2 * mc,), K_B)
    for i in range(2, 7):
        _attention_block(spec, f"diffusion.contextual_embedder.{i}", 2 * mc, heads)
    for i in range(3):
        _diffusion_layer(spec, f"diffusion.conditioning_timestep_integrator.{i}", mc, heads)
    spec["diffusion.integrating_conv.weight"] = ((mc, 2 * mc, 1), K_W)
    spec["diffusion.integrating_conv.bias"] = ((mc,), K_B)
    for i in range(d["num_layers"]):
        _diffusion_layer(spec, f"diffusion.layers.{i}", mc, heads)
    for i in range(d["num_layers"], d["num_layers"] + 3):
        _diff_resblock(spec, f"diffusion.layers.{i}", mc)
    spec["diffusion.out.0.weight"] = ((mc,), K_NG)
    spec["diffusion.out.0.bias"] = ((mc,), K_NB)
    spec["diffusion.out.2.weight"] = ((d["out_channels"], mc, 3), K_W)
    spec["diffusion.out.2.bias"] = ((d["out_channels"],), K_B)

    # ---- in_proj + enc_p (SpecEncoder, vqvae/model_24k.py:71-124, 588-590)
    hid, filt, nh, nl, ks = v["hidden_channels"], v["filter_channels"], v["n_heads"], v["n_layers"], v["kernel_size"]
    spec["in_proj.weight"] = ((inter, n_mel, 3), K_W)
    spec["in_proj.bias"] = ((inter,), K_B)
    for i in range(nl):
        p = f"enc_p.encoder.attn_layers.{i}"
        spec[p + ".emb_rel_k"] = ((1, 9, hid // nh), K_RELE)
        spec[p + ".emb_rel_v"] = ((1, 9, hid // nh), K_RELE)
        for n in ("conv_q", "conv_k", "conv_v", "conv_o"):
            spec[p + f".{n}.weight"] = ((hid, hid, 1), K_W)
            spec[p + f".{n}.bias"] = ((hid,), K_B)
    for i in range(nl):
        spec[f"enc_p.encoder.norm_layers_1.{i}.gamma"] = ((hid,), K_NG)
        spec[f"enc_p.encoder.norm_layers_1.{i}.beta"] = ((hid,), K_NB)
    for i in range(nl):
        spec[f"enc_p.encoder.ffn_layers.{i}.conv_1.weight"] = ((filt, hid, ks), K_W)
        spec[f"enc_p.encoder.ffn_layers.{i}.conv_1.bias"] = ((filt,), K_B)
        spec[f"enc_p.encoder.ffn_layers.{i}.conv_2.weight"] = ((hid, filt, ks), K_W)
        spec[f"enc_p.encoder.ffn_layers.{i}.conv_2.bias"] = ((hid,), K_B)
    for i in range(nl):
        spec[f"enc_p.encoder.norm_layers_2.{i}.gamma"] = ((hid,), K_NG)
        spec[f"enc_p.encoder.norm_layers_2.{i}.beta"] = ((hid,), K_NB)
    spec["enc_p.out_proj.weight"] = ((inter, hid, 1), K_W)
    spec["enc_p.out_proj.bias"] = ((inter,), K_B)
    spec["enc_p.proj.weight"] = ((2 * inter, inter, 1), K_W)
    spec["enc_p.proj.bias"] = ((2 * inter,), K_B)

    # ---- flow (ResidualCouplingBlock, vqvae/model_24k.py:127-169; flows 1,3,5,7 are Flip)
    half = inter // 2
    for f in (0, 2, 4, 6):
        p = f"flow.flows.{f}"
        spec[p + ".pre.weight"] = ((hid, half, 1), K_W)
        spec[p + ".pre.bias"] = ((hid,), K_B)
        for l in range(4):
            spec[p + f".enc.in_layers.{l}.bias"] = ((2 * hid,), K_B)
            spec[p + f".enc.in_layers.{l}.weight_g"] = ((2 * hid, 1, 1), K_WN_G)
            spec[p + f".enc.in_layers.{l}.weight_v"] = ((2 * hid, hid, 5), K_WN_V)
        for l in range(4):
            rs = 2 * hid if l < 3 else hid
            spec[p + f".enc.res_skip_layers.{l}.bias"] = ((rs,), K_B)
            spec[p + f".enc.res_skip_layers.{l}.weight_g"] = ((rs, 1, 1), K_WN_G)
            spec[p + f".enc.res_skip_layers.{l}.weight_v"] = ((rs, hid, 1), K_WN_V)
        spec[p + ".enc.cond_layer.bias"] = ((2 * hid * 4,), K_B)
        spec[p + ".enc.cond_layer.weight_g"] = ((2 * hid * 4, 1, 1), K_WN_G)
        spec[p + ".enc.cond_layer.weight_v"] = ((2 * hid * 4, gin, 1), K_WN_V)
        spec[p + ".post.weight"] = ((half, hid, 1), K_W)
        spec[p + ".post.bias"] = ((half,), K_B)

    # ---- ref_enc (MelStyleEncoder hidden 128, vqvae/model_24k.py:594-596)
    _mel_style_encoder(spec, "ref_enc", n_mel, 128, gin)

    # ---- gpt (UnifiedVoice, gpt/model.py:265-331)
    md = g["model_dim"]
    _mel_style_encoder(spec, "gpt.conditioning_encoder", g["spec_channels"], md // 2, md)
    spec["gpt.text_embedding.weight"] = ((g["number_text_tokens"] + 1, md), K_EMB)
    spec["gpt.mel_embedding.weight"] = ((g["number_mel_codes"], md), K_EMB)
    for l in range(g["layers"]):
        p = f"gpt.gpt.h.{l}"
        spec[p + ".ln_1.weight"] = ((md,), K_NG)
        spec[p + ".ln_1.bias"] = ((md,), K_NB)
        spec[p + ".attn.c_attn.weight"] = ((md, 3 * md), K_HFW)
        spec[p + ".attn.c_attn.bias"] = ((3 * md,), K_HFB)
        spec[p + ".attn.c_proj.weight"] = ((md, md), K_HFW)
        spec[p + ".attn.c_proj.bias"] = ((md,), K_HFB)
        spec[p + ".ln_2.weight"] = ((md,), K_NG)
        spec[p + ".ln_2.bias"] = ((md,), K_NB)
        spec[p + ".mlp.c_fc.weight"] = ((md, 4 * md), K_HFW)
        spec[p + ".mlp.c_fc.bias"] = ((4 * md,), K_HFB)
        spec[p + ".mlp.c_proj.weight"] = ((4 * md, md), K_HFW)
        spec[p + ".mlp.c_proj.bias"] = ((md,), K_HFB)
    spec["gpt.gpt.ln_f.weight"] = ((md,), K_NG)
    spec["gpt.gpt.ln_f.bias"] = ((md,), K_NB)
    # max_mel_tokens + 2 + max_conditioning_inputs(1) ; max_text_tokens + 2   (gpt/model.py:313-314)
    spec["gpt.mel_pos_embedding.emb.weight"] = ((g["max_mel_tokens"] + 3, md), K_EMB)
    spec["gpt.text_pos_embedding.emb.weight"] = ((g["max_text_tokens"] + 2, md), K_EMB)
    spec["gpt.final_norm.weight"] = ((md,), K_NG)
    spec["gpt.final_norm.bias"] = ((md,), K_NB)
    spec["gpt.mel_head.weight"] = ((g["number_mel_codes"], md), K_W)
    spec["gpt.mel_head.bias"] = ((g["number_mel_codes"],), K_B)

    # ---- VQ decode path of infer_gpt / infer_vqvae (vqvae/model_24k.py:811-847, 610-624)
    spec["quantizer.vq.layers.0._codebook.embed"] = ((v["vq_bins"], 8), K_UNCOND)
    spec["quantizer.vq.layers.0.project_out.weight"] = ((4 * inter, 8), K_W)
    spec["quantizer.vq.layers.0.project_out.bias"] = ((4 * inter,), K_B)
    spec["quantizer.vq.layers.0.project_in.weight"] = ((8, 4 * inter), K_W)
    spec["quantizer.vq.layers.0.project_in.bias"] = ((8,), K_B)
    spec["vq_enc.1.weight"] = ((n_mel,), K_NG)
    spec["vq_enc.1.bias"] = ((n_mel,), K_NB)
    spec["vq_enc.3.weight"] = ((2 * inter, n_mel, 3), K_W)
    spec["vq_enc.3.bias"] = ((2 * inter,), K_B)
    spec["vq_enc.5.weight"] = ((4 * inter, 2 * inter, 3), K_W)
    spec["vq_enc.5.bias"] = ((4 * inter,), K_B)
    spec["vq_enc.7.weight"] = ((4 * inter, 4 * inter, 3), K_W)
    spec["vq_enc.7.bias"] = ((4 * inter,), K_B)
    spec["vq_dec.1.weight"] = ((4 * inter,), K_NG)
    spec["vq_dec.1.bias"] = ((4 * inter,), K_NB)
    spec["vq_dec.3.weight"] = ((4 * inter, 2 * inter, 3), K_W)            # ConvTranspose1d [in, out, k]
    spec["vq_dec.3.bias"] = ((2 * inter,), K_B)
    spec["vq_dec.5.weight"] = ((2 * inter, inter, 3), K_W)
    spec["vq_dec.5.bias"] = ((inter,), K_B)
    spec["vq_dec.7.weight"] = ((n_mel, inter, 3), K_W)
    spec["vq_dec.7.bias"] = ((n_mel,), K_B)
    _mel_style_encoder(spec, "vq_ref_enc", n_mel, 128, 4 * inter)
    return spec


def _rng_for(seed: int, name: str) -> np.random.Generator:
    return np.random.Generator(np.random.Philox(key=[int(seed) & 0xFFFFFFFFFFFFFFFF, zlib.crc32(name.encode())]))


def synthetic_state_dict(seed: int = 0, cfg=None, only_prefixes=None, variant=None) -> "OrderedDict[str, np.ndarray]":
    """Deterministic random-init weights in *state-dict form* (fp32 numpy).

    ``variant="signal"``: the same draws, rescaled so that the vocoder's output DEPENDS ON ITS INPUT.  With the plain fan-in
    init every generator conv attenuates its input by ~0.58 (uniform +-1/sqrt(fan_in) has std 1/sqrt(3 fan_in)) while every bias
    injects +-0.05, so after 7 convs the waveform is 99 % bias (generator(z) - generator(0) is 1e-3 RMS on a 7e-3 RMS output): a
    waveform comparison under those weights cannot see an error upstream.  The variant multiplies the generator's non-residual
    gains (conv_pre, cond, ups.*.weight_g, conv_post) by 2.0 and its biases by 0.3 (output RMS 0.2, 6/7 of it driven by z, tanh
    not saturated) and makes the prior's mean count against its noise (enc_p.proj: m_p rows x 6, logs_p bias - 1.5).

    Every tensor is drawn from its own Philox stream keyed by (seed, crc32(name)),
    so a subset (``only_prefixes``) yields the same values as the full set.  Zero-
    initialised tensors of the reference (AttentionBlock.proj_out,
    vqvae/utils/diff_util.py:203; ResidualCouplingLayer.post,
    vqvae/modules/modules.py:453-454) are given non-zero values so the maths they
    gate is exercised.  Scales are fan-in based so activations stay O(1).
    """
    spec = inference_param_spec(cfg)
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, (shape, kind) in spec.items():
        if only_prefixes is not None and not name.startswith(tuple(only_prefixes)):
            continue
        r = _rng_for(seed, name)
        if kind in (K_W, K_WN_V):
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
            if name.startswith("dec.ups.") and kind == K_WN_V:
                # ConvTranspose1d [in, out, k]: each output sample sums in*k/stride taps
                fan_in = shape[0] * max(1, shape[2] // 2)
            b = 1.0 / np.sqrt(fan_in)
            a = r.uniform(-b, b, size=shape)
        elif kind == K_B:
            a = r.uniform(-0.05, 0.05, size=shape)
        elif kind == K_NG:
            a = r.uniform(0.8, 1.2, size=shape)
        elif kind == K_NB:
            a = r.uniform(-0.1, 0.1, size=shape)
        elif kind in (K_EMB, K_HFW, K_HFB):
            a = r.normal(0.0, 0.02, size=shape)
        elif kind == K_WN_G:
            # magnitude = ||v|| * U(.7,1.3): needs v, drawn from v's own stream
            vname = name[: -len("weight_g")] + "weight_v"
            vshape = spec[vname][0]
            v = synthetic_tensor(seed, vname, spec)
            nrm = np.sqrt((v.astype(np.float64) ** 2).reshape(vshape[0], -1).sum(1)).reshape(shape)
            a = nrm * r.uniform(0.7, 1.3, size=shape)
        elif kind == K_RELB:
            a = r.normal(0.0, 0.5, size=shape)
        elif kind == K_RELE:
            a = r.normal(0.0, shape[-1] ** -0.5, size=shape)
        elif kind == K_UNCOND:
            a = r.normal(0.0, 1.0, size=shape)
        else:  # pragma: no cover
            raise ValueError(kind)
        if variant == "signal":
            a = _signal_variant(name, a)
        elif variant is not None:
            raise ValueError(f"unknown weight variant {variant!r}")
        out[name] = np.ascontiguousarray(a, dtype=np.float32)
    return out


def _signal_variant(name, a):
    if name.startswith("dec."):
        if name.endswith("bias"):
            return a * 0.3
        if name in ("dec.conv_pre.weight", "dec.conv_post.weight", "dec.cond.weight") or (name.startswith("dec.ups.") and name.endswith("weight_g")):
            return a * 2.0
    if name == "enc_p.proj.weight":
        a = a.copy()
        a[: a.shape[0] // 2] *= 6.0
    if name == "enc_p.proj.bias":
        a = a.copy()
        a[a.shape[0] // 2:] -= 1.5
    return a


def synthetic_tensor(seed, name, spec=None):
    spec = spec or inference_param_spec()
    shape, kind = spec[name]
    assert kind in (K_W, K_WN_V)
    r = _rng_for(seed, name)
    fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
    if name.startswith("dec.ups.") and kind == K_WN_V:
        fan_in = shape[0] * max(1, shape[2] // 2)
    b = 1.0 / np.sqrt(fan_in)
    return np.ascontiguousarray(r.uniform(-b, b, size=shape), dtype=np.float32)


def fold_weight_norm(state: dict) -> "OrderedDict[str, np.ndarray]":
    """state-dict form -> folded form: `X.weight_g`,`X.weight_v` -> `X.weight`.

    torch.nn.utils.weight_norm(dim=0): w = g * v / ||v||, the norm taken over
    every dim except 0 (for ConvTranspose1d dim 0 is the *input* channel,
    SURVEY.md §5).  Also accepts the new-style parametrization keys
    (`parametrizations.weight.original0/1`).  Unneeded keys are passed through.
    """
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for k, val in state.items():
        a = _np(val)
        if k.endswith(".weight_v") or k.endswith(".parametrizations.weight.original1"):
            continue
        if k.endswith(".weight_g") or k.endswith(".parametrizations.weight.original0"):
            if k.endswith(".weight_g"):
                base = k[: -len(".weight_g")]
                v = _np(state[base + ".weight_v"])
            else:
                base = k[: -len(".parametrizations.weight.original0")]
                v = _np(state[base + ".parametrizations.weight.original1"])
            v64 = v.astype(np.float64)
            nrm = np.sqrt((v64 ** 2).reshape(v.shape[0], -1).sum(1)).reshape((-1,) + (1,) * (v.ndim - 1))
            out[base + ".weight"] = (a.astype(np.float64).reshape(nrm.shape) * v64 / nrm).astype(np.float32)
            continue
        out[k] = a
    return out


def _np(x):
    if isinstance(x, np.ndarray):
        return x
    # torch tensor without importing torch here
    return x.detach().cpu().float().numpy() if hasattr(x, "detach") else np.asarray(x)


def folded_param_names(cfg=None):
    names = []
    for k in inference_param_spec(cfg):
        if k.endswith(".weight_v"):
            continue
        names.append(k[: -len("_g")] if k.endswith(".weight_g") else k)
    return names


def select_inference_params(state: dict, cfg=None) -> "OrderedDict[str, np.ndarray]":
    """Fold + keep exactly the tensors the hot path needs; raise on missing/mis-shaped."""
    folded = fold_weight_norm({k: v for k, v in state.items() if not k.startswith("gpt.inference_model.")})
    spec = inference_param_spec(cfg)
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for k, (shape, kind) in spec.items():
        if k.endswith(".weight_v"):
            continue
        if k.endswith(".weight_g"):
            k = k[: -len("_g")]
            shape = spec[k + "_v"][0]
        if k not in folded:
            raise KeyError(f"checkpoint is missing '{k}'")
        a = np.ascontiguousarray(_np(folded[k]), dtype=np.float32)
        if tuple(a.shape) != tuple(shape):
            raise ValueError(f"'{k}': expected shape {tuple(shape)}, got {tuple(a.shape)}")
        out[k] = a
    return out
