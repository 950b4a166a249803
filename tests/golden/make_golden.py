#!/usr/bin/env python3
"""Generate the golden fixtures in this directory FROM THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference, CPU only).  It imports
the reference's Python under the shim recorded in SURVEY.md App. C, loads the
deterministic synthetic weights of detail_tts_amd.weights.synthetic_state_dict
(seed 0) into the reference's own `SynthesizerTrn`, drives the reference's own
functions stage by stage on small seeded inputs, and stores ONLY inputs and
expected outputs (data) as .npz files.  Weights are not stored: tests regenerate
them from the seed.  The reference's four RNG sites are patched to draw from the
Philox noise spec of oracle/philox.py (SURVEY.md §7 "Parity through randomness").

    python tests/golden/make_golden.py            # writes tests/golden/*.npz
"""
import contextlib
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

REF = "/root/reference"
SEED_W = 0          # weight seed
SEED_N = 1234       # noise seed


def install_shim():
    from transformers import GPT2PreTrainedModel, GenerationMixin, LogitsProcessor, GPT2Model, GPT2Config  # noqa: F401
    sys.modules["transformers"].LogitsWarper = LogitsProcessor

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Dummy:
        def __init__(self, *a, **k):
            pass

    stub("transformers.utils.model_parallel_utils", get_device_map=lambda *a, **k: None, assert_device_map=lambda *a, **k: None)
    ta = stub("torchaudio")
    ta.transforms = stub("torchaudio.transforms", MelSpectrogram=_Dummy, Resample=_Dummy)
    ta.functional = stub("torchaudio.functional")
    kd = stub("k_diffusion")
    kd.sampling = stub("k_diffusion.sampling", sample_dpmpp_2m=None, sample_euler_ancestral=None)
    lb = stub("librosa")
    lb.util = stub("librosa.util", normalize=None, pad_center=None, tiny=None)
    lb.filters = stub("librosa.filters", mel=None)
    stub("pypinyin", lazy_pinyin=None, Style=None)
    sys.path.insert(0, REF)
    import gpt.model as gm
    gm.GPT2InferenceModel.__bases__ = (GPT2PreTrainedModel, GenerationMixin)


def build_reference_model(variant=None):
    """variant: detail_tts_amd.weights.synthetic_state_dict's `variant` (None = the plain seed-0 set, "signal" = the vocoder rescaled so
    that its output depends on its input)"""
    import torch
    from vqvae.utils.data_utils import HParams
    from vqvae.model_24k import SynthesizerTrn
    from detail_tts_amd.weights import synthetic_state_dict
    cfg = json.load(open(os.path.join(REF, "vqvae/configs/config_24k.json")))
    cfg["diffusion"].pop("g_channels")
    hps = HParams(**cfg)
    torch.manual_seed(0)
    model = SynthesizerTrn(1024 // 2 + 1, 10240 // 256, **hps.vaegan, cfg=hps).eval()
    sd = {k: torch.from_numpy(v) for k, v in synthetic_state_dict(SEED_W, variant=variant).items()}
    res = model.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    allowed = ("enc_q.", "quantizer.", "vq_enc.", "vq_dec.", "vq_ref_enc.", "gpt.text_head.",
               "diffusion.code_embedding.", "diffusion.code_converter.", "diffusion.mel_head.",
               "gpt.inference_model.", "gpt.gpt.wte.")
    bad = [k for k in res.missing_keys if not k.startswith(allowed)]
    assert not bad, bad
    # a trained checkpoint carries `inited` = 1; without it the first quantizer forward would k-means re-initialise the codebook
    model.quantizer.vq.layers[0]._codebook.inited.fill_(1.0)
    return model


@contextlib.contextmanager
def philox_rng(sample_id=0):
    """Patch the reference's RNG sites to the Philox spec (oracle/philox.py)."""
    import torch
    from oracle import philox
    state = {"diff_step": None, "gpt_step": 0}
    o_randn, o_randn_like, o_multi = torch.randn, torch.randn_like, torch.multinomial

    def randn(*shape, **kw):                       # vqvae/model_24k.py:488
        shape = tuple(shape[0]) if len(shape) == 1 and isinstance(shape[0], (tuple, list)) else tuple(shape)
        n = int(np.prod(shape))
        state["diff_step"] = 49
        return torch.from_numpy(philox.normal(SEED_N, sample_id, philox.STAGE_DIFF_INIT, 0, n).reshape(shape))

    def randn_like(x, **kw):
        n = x.numel()
        if x.shape[1] == 128:                      # vqvae/utils/diffusion.py:480
            i = state["diff_step"]
            state["diff_step"] = i - 1
            z = philox.normal(SEED_N, sample_id, philox.STAGE_DIFF_STEP, i, n)
        else:                                      # vqvae/model_24k.py:860
            z = philox.normal(SEED_N, sample_id, philox.STAGE_FLOW_PRIOR, 0, n)
        return torch.from_numpy(z.reshape(tuple(x.shape)))

    def multinomial(probs, num_samples, **kw):     # HF GenerationMixin._sample
        assert num_samples == 1
        out = []
        for b in range(probs.shape[0]):
            u = philox.uniform_scalar(SEED_N, sample_id + b, philox.STAGE_GPT_SAMPLE, state["gpt_step"])
            c = np.cumsum(probs[b].double().numpy())
            out.append(min(int(np.searchsorted(c, u * c[-1], side="right")), probs.shape[1] - 1))
        state["gpt_step"] += 1
        return torch.tensor(out, dtype=torch.long)[:, None]

    torch.randn, torch.randn_like, torch.multinomial = randn, randn_like, multinomial
    try:
        yield state
    finally:
        torch.randn, torch.randn_like, torch.multinomial = o_randn, o_randn_like, o_multi


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        a = np.asarray(v.detach().numpy() if hasattr(v, "detach") else v)
        if a.dtype == np.float64 and not k.startswith("f64_"):
            a = a.astype(np.float32)
        out[k] = a
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB  " + ", ".join(f"{k}{tuple(v.shape)}" for k, v in out.items()))


def main():
    install_shim()
    import torch
    import vqvae.modules.commons as commons
    from vqvae.model_24k import do_spectrogram_diffusion, denormalize_torch_mel
    from vqvae.utils.xtransformers import RelativePositionBias
    torch.set_grad_enabled(False)
    m = build_reference_model()
    rs = np.random.RandomState(1)

    T_ref, L0, n = 64, 12, 12
    T = 4 * n
    refer = (rs.randn(1, 128, T_ref) * 2 - 5).astype(np.float32)
    text = np.concatenate([rs.randint(3, 255, (1, L0)), [[0]]], 1).astype(np.int32)   # api.py:24-25
    codes = np.random.RandomState(2).randint(0, 8192, (1, n)).astype(np.int64)
    refer_t, text_t, codes_t = torch.from_numpy(refer), torch.from_numpy(text), torch.from_numpy(codes)
    rl = torch.tensor([T_ref])

    # ---- 0. relative position buckets + schedule tables ---------------------------------
    rel = torch.arange(-2100, 2101)
    save("rel_bucket", rel=rel, bucket=RelativePositionBias._relative_position_bucket(rel, causal=False, num_buckets=32, max_distance=64))
    d = m.infer_diffuser
    save("schedule", timestep_map=np.array(d.timestep_map), f64_betas=d.betas,
         f64_sqrt_recip_alphas_cumprod=d.sqrt_recip_alphas_cumprod, f64_sqrt_recipm1_alphas_cumprod=d.sqrt_recipm1_alphas_cumprod,
         f64_posterior_log_variance_clipped=d.posterior_log_variance_clipped, f64_posterior_mean_coef1=d.posterior_mean_coef1,
         f64_posterior_mean_coef2=d.posterior_mean_coef2, num_timesteps=np.array(d.num_timesteps))

    # ---- 1. MelStyleEncoder (A3, A17), with and without padding --------------------------
    x2 = (rs.randn(2, 128, 40) * 2 - 5).astype(np.float32)
    len2 = np.array([40, 29])
    mask2 = commons.sequence_mask(torch.from_numpy(len2), 40).unsqueeze(1).float()
    save("mel_style", refer=refer, x2=x2, len2=len2,
         gpt_cond=m.gpt.conditioning_encoder(refer_t, commons.sequence_mask(rl, T_ref).unsqueeze(1).float()),
         gpt_cond2=m.gpt.conditioning_encoder(torch.from_numpy(x2), mask2),
         ref_enc2=m.ref_enc(torch.from_numpy(x2) * mask2, mask2))

    # ---- 2. GPT: prefix, uncached logits over a forced history, latents (A2,A4-A6,A8) ---
    g = m.gpt
    import torch.nn.functional as F
    t_in = F.pad(text_t, (0, 1), value=g.stop_text_token)
    t_in, _ = g.build_aligned_inputs_and_targets(t_in, g.start_text_token, g.stop_text_token)
    temb = g.text_embedding(t_in) + g.text_pos_embedding(t_in)
    cond = g.conditioning_encoder(refer_t, commons.sequence_mask(rl, T_ref).unsqueeze(1).float()).transpose(1, 2)
    prefix = torch.cat([cond, temb], 1)
    g.inference_model.store_mel_emb(prefix)
    Pn = prefix.shape[1]
    ids = torch.cat([torch.ones(1, Pn, dtype=torch.long), torch.tensor([[g.start_mel_token]]), codes_t], 1)
    logits_full = g.inference_model(input_ids=ids, return_dict=True).logits          # [1, Pn+1+n, 8194]
    latent = g(refer_t, rl, text_t, torch.tensor([text.shape[1]]), codes_t.clone(), torch.tensor([n * 1024]), return_latent=True)
    save("gpt_forced", refer=refer, text=text, codes=codes, prefix=prefix,
         logits_steps=np.array([0, 5, n]), logits=logits_full[0, [Pn + 0, Pn + 5, Pn + n]], latent=latent)

    # ---- 3. sampler: HF processors on logits fixtures (A7) ------------------------------
    from transformers import LogitsProcessorList, RepetitionPenaltyLogitsProcessor, TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper
    sc = logits_full[0, [Pn + 0, Pn + 5, Pn + n]].clone() * 20.0          # widen the distribution of random-weight logits
    hist = torch.cat([torch.ones(3, 4, dtype=torch.long), torch.full((3, 1), 8192), codes_t[:, :8].repeat(3, 1)], 1)
    outs = {}
    for tag, tk in (("none", None), ("k50", 50)):
        procs = [RepetitionPenaltyLogitsProcessor(2.0), TemperatureLogitsWarper(0.8)]
        if tk:
            procs.append(TopKLogitsWarper(tk))
        procs.append(TopPLogitsWarper(0.8))
        outs["filtered_" + tag] = LogitsProcessorList(procs)(hist, sc.clone())
    save("sampler_filter", scores=sc, history=hist, **outs)

    # ---- 4. full HF generate under the Philox multinomial (A2+A5+A7 loop semantics) -----
    with philox_rng(sample_id=7):
        gen = g.inference_speech_tortoise(refer_t, rl, text_t, do_sample=True, top_p=0.8, temperature=0.8,
                                          num_return_sequences=1, length_penalty=1.0, repetition_penalty=2.0,
                                          max_generate_length=10)
    save("gpt_generate", refer=refer, text=text, sample_id=np.array(7), seed=np.array(SEED_N), codes=gen)

    # ---- 5. diffusion conditioning (A9, A10) --------------------------------------------
    cond_lat = m.diffusion.get_conditioning(refer_t)
    code_emb = m.diffusion.timestep_independent(latent, cond_lat, T, False)
    save("diff_cond", refer=refer, latent=latent, cond_latent=cond_lat, code_emb=code_emb)

    # ---- 6. one DiffusionTts.forward, cond + uncond (A13, A14) ---------------------------
    x = rs.randn(1, 128, T).astype(np.float32)
    ts = torch.tensor([3836])
    save("diff_forward", x=x, ts=ts, code_emb=code_emb,
         out_cond=m.diffusion(torch.from_numpy(x), ts, precomputed_aligned_embeddings=code_emb),
         out_uncond=m.diffusion(torch.from_numpy(x), ts, precomputed_aligned_embeddings=code_emb, conditioning_free=True))

    # ---- 7. first 3 sampler steps + the last one (A11, A12) ------------------------------
    tr = {}
    with philox_rng(sample_id=0):
        img = torch.randn((1, 128, T))
        tr["x_init"] = img
        for i in (49, 48, 47):
            out = d.p_sample(m.diffusion, img, torch.tensor([i]), model_kwargs={"precomputed_aligned_embeddings": code_emb})
            img = out["sample"]
            tr[f"x_after_{i}"] = img
            tr[f"x0_after_{i}"] = out["pred_xstart"]
    with philox_rng(sample_id=0) as st:
        st["diff_step"] = 0
        x_last = torch.from_numpy((rs.randn(1, 128, T) * 0.5).astype(np.float32))
        out = d.p_sample(m.diffusion, x_last, torch.tensor([0]), model_kwargs={"precomputed_aligned_embeddings": code_emb})
        tr["x_before_0"], tr["x_after_0"] = x_last, out["sample"]
    save("diff_sampler_steps", code_emb=code_emb, seed=np.array(SEED_N), sample_id=np.array(0), **tr)

    # ---- 8. stage C pieces (A16-A20) -----------------------------------------------------
    mel = (rs.randn(1, 128, T) * 2 - 5).astype(np.float32)
    mel_t = torch.from_numpy(mel)
    yl = torch.tensor([T])
    ymask = commons.sequence_mask(yl, T).unsqueeze(1).float()
    gref = m.ref_enc(mel_t * ymask, ymask)
    xin = m.in_proj(mel_t)
    _, m_p, logs_p = m.enc_p(xin, yl)
    with philox_rng(sample_id=3):
        z_p = m_p + torch.randn_like(m_p) * torch.exp(logs_p) * 0.667
    z = m.flow(z_p, ymask, g=gref, reverse=True)
    wav = m.dec(z, g=gref)
    with philox_rng(sample_id=3):
        wav2 = m.infer_flowvae(mel_t, yl, None)
    assert torch.equal(wav, wav2)
    save("vocoder", mel=mel, g=gref, m_p=m_p, logs_p=logs_p, z_p=z_p, z=z, wav=wav, seed=np.array(SEED_N), sample_id=np.array(3))

    # ---- 9. end-to-end with forced codes + Philox noise (A1) -----------------------------
    orig = g.inference_speech_tortoise
    g.inference_speech_tortoise = lambda *a, **k: torch.cat([codes_t, torch.tensor([[g.stop_mel_token]])], 1)
    with philox_rng(sample_id=5):
        wav_e2e = m.infer(text_t, torch.tensor([text.shape[1]]), refer_t, rl)
    g.inference_speech_tortoise = orig
    save("e2e_forced", refer=refer, text=text, codes=codes, wav=wav_e2e, seed=np.array(SEED_N), sample_id=np.array(5))

    # ---- 10. infer_gpt's VQ decode path (vqvae/model_24k.py:811-847) with forced codes -----------------------
    vq_codes = rs.randint(0, 8192, size=(1, 9)).astype(np.int64)
    vq_codes_t = torch.from_numpy(vq_codes)
    rmask = commons.sequence_mask(rl, refer_t.size(2)).unsqueeze(1).float()
    latent = m.quantizer.decode(vq_codes_t.unsqueeze(0))
    g_vq = m.vq_ref_enc(refer_t * rmask, rmask)
    recon = m.vq_dec(latent + g_vq)
    g.inference_speech_tortoise = lambda *a, **k: torch.cat([vq_codes_t, torch.tensor([[g.stop_mel_token]])], 1)
    with philox_rng(sample_id=6):
        wav_vq = m.infer_gpt(text_t, torch.tensor([text.shape[1]]), refer_t, rl)
    g.inference_speech_tortoise = orig
    save("vq_path", refer=refer, codes=vq_codes, latent=latent, g_vq=g_vq, recon=recon, wav=wav_vq, seed=np.array(SEED_N),
         sample_id=np.array(6))
    # encode side (vqvae/model_24k.py:877-880)
    enc_codes, enc_xvq = m.encode(refer_t, rl)
    save("vq_encode", mel=refer, codes=enc_codes, x_vq=enc_xvq)

    # ---- 12 (host). text front-end: the reference tokenizer on its own vocabulary (bpe_tokenizers/voice_tokenizer.py:31-54); the
    # pinyin sentence is the one pinned in demo.ipynb (api.py:14 through pypinyin)
    from bpe_tokenizers.voice_tokenizer import VoiceBpeTokenizer as RefTok
    tok = RefTok(os.path.join(REF, "bpe_tokenizers/zh_tokenizer.json"))
    kat_texts = [" da4 jia1 hao3 \uff0c jin1 tian1 lai2 dian3 da4 jia1 xiang3 kan4 de5 dong1 xi1 \u3002 ",
                 " ni3 hao3 {shi4 jie4} [ce4 shi4] \u2014 yi1 er4 san1 ! ", "hello world, this is a test."]
    with open(os.path.join(HERE, "tokenizer_kat.json"), "w") as fh:
        json.dump([{"text": t, "ids": tok.encode(t), "decoded": tok.decode(np.array(tok.encode(t)))} for t in kat_texts], fh, indent=1)
    print("tokenizer_kat:", [len(tok.encode(t)) for t in kat_texts])

    # ---- 11. prompt front-end (api.py:40-45, vqvae/utils/data_utils.py:56-155): the reference's own STFT / magnitude / log
    # arithmetic.  librosa is absent here, so the mel filterbank handed to the reference function is the oracle's restatement of
    # librosa.filters.mel (parity for the filterbank values themselves stays unpinned); torchaudio's resampler is not exercised.
    import vqvae.utils.data_utils as du
    from oracle import frontend as FE
    du.librosa_mel_fn = lambda sr, n_fft, n_mels, fmin, fmax: FE.mel_filterbank(sr, n_fft, n_mels, fmin, fmax)
    tt = np.arange(13000) / 24000.0
    yw = (0.3 * np.sin(2 * np.pi * (200 + 4000 * tt) * tt) + 0.1 * rs.randn(13000) * np.exp(-3 * tt) + 0.02 * rs.randn(13000)).astype(np.float32)
    yw = np.stack([yw, np.roll(yw, 777) * 0.5]).astype(np.float32)
    yw_t = torch.from_numpy(yw)
    spec_lin = du.spectrogram_torch(yw_t, 1024, 256, 1024)
    mel_ref = du.mel_spectrogram_torch(yw_t, 1024, 128, 24000, 256, 1024, 0.0, None)
    save("frontend", wav=yw, spec=spec_lin, mel=mel_ref)


if __name__ == "__main__":
    main()
