"""The numpy oracle vs outputs of the reference itself (tests/golden/*.npz, made by
tests/golden/make_golden.py from /root/reference).  This is what pins the oracle."""
import numpy as np
import pytest

from oracle import diffusion as D
from oracle import gpt as G
from oracle import philox, vocoder as V
from oracle import pipeline


def maxabs(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))))


def test_philox_known_answers():
    # Random123 kat_vectors for philox4x32-10
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, exp in kat:
        out = philox.philox4x32(*ctr, *key)
        assert tuple(int(o) for o in out) == exp
    z = philox.normal(1234, 3, philox.STAGE_DIFF_STEP, 17, 200001)
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01 and np.isfinite(z).all()


def test_rel_bucket(golden):
    g = golden("rel_bucket")
    assert np.array_equal(D.rel_bucket(g["rel"]), g["bucket"])


def test_schedule(golden):
    g = golden("schedule")
    s = D.make_schedule()
    assert np.array_equal(s["timestep_map"], g["timestep_map"])
    assert s["num_timesteps"] == int(g["num_timesteps"]) == 50
    for k in ("betas", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_log_variance_clipped",
              "posterior_mean_coef1", "posterior_mean_coef2"):
        np.testing.assert_allclose(s[k], g["f64_" + k], rtol=1e-13, atol=0)


def test_mel_style_encoder(weights, golden):
    g = golden("mel_style")
    assert maxabs(G.mel_style_encoder(weights, "gpt.conditioning_encoder", g["refer"], [64]), g["gpt_cond"]) < 1e-5
    assert maxabs(G.mel_style_encoder(weights, "gpt.conditioning_encoder", g["x2"], g["len2"]), g["gpt_cond2"]) < 1e-5
    mf = (np.arange(40)[None, :] < g["len2"][:, None])[:, None, :].astype(np.float32)
    assert maxabs(G.mel_style_encoder(weights, "ref_enc", g["x2"] * mf, g["len2"]), g["ref_enc2"]) < 1e-5


def test_gpt_prefix_logits_latents(weights, golden):
    g = golden("gpt_forced")
    prefix = G.prefix_embeddings(weights, g["refer"], [g["refer"].shape[2]], g["text"])
    assert prefix.shape == g["prefix"].shape and maxabs(prefix, g["prefix"]) < 1e-5
    mel_ids = np.concatenate([[[G.START_MEL]], g["codes"]], 1)
    logits, lat = G.logits_nocache(weights, prefix, mel_ids)
    assert maxabs(logits[0, g["logits_steps"]], g["logits"]) < 2e-5
    latent = G.latents_teacher_forced(weights, g["refer"], [g["refer"].shape[2]], g["text"], g["codes"])
    assert maxabs(latent, g["latent"]) < 2e-5
    # decode-time hidden states == teacher-forced latents
    assert maxabs(lat[:, :-1], g["latent"]) < 2e-5


@pytest.mark.parametrize("tag,top_k", [("none", None), ("k50", 50)])
def test_sampler_filter(golden, tag, top_k):
    g = golden("sampler_filter")
    for r in range(3):
        f = G.process_logits(g["scores"][r], g["history"][r], top_k=top_k)
        ref = g["filtered_" + tag][r]
        assert np.array_equal(np.isfinite(f), np.isfinite(ref))
        keep = np.isfinite(ref)
        assert maxabs(f[keep], ref[keep]) < 1e-5


def test_generate_loop(weights, golden):
    g = golden("gpt_generate")
    for cache in (False, True):
        codes = G.generate(weights, g["refer"], [g["refer"].shape[2]], g["text"], int(g["seed"]), [int(g["sample_id"])],
                           max_generate_length=10, top_k=50, use_cache=cache)
        assert np.array_equal(codes, g["codes"]), (cache, codes, g["codes"])


def test_generate_off_path_branches(weights, golden):
    """inference_speech_tortoise's branches SynthesizerTrn.infer never takes (gpt/model.py:533-544): greedy search, num_return_sequences = 2
    (HF repeat_interleave: row r of the expanded batch draws from noise stream sample_id + r), input_tokens, typical sampling - against the reference's own
    HF generate (tests/golden/make_golden_r5.py)."""
    g = golden("gpt_generate_branches")
    Tr, sid, seed = g["refer"].shape[2], int(g["sample_id"]), int(g["seed"])
    codes = G.generate(weights, g["refer"], [Tr], g["text"], seed, [sid], max_generate_length=10, do_sample=False)
    assert np.array_equal(codes, g["greedy"]), (codes, g["greedy"])
    codes = G.generate(weights, np.repeat(g["refer"], 2, 0), [Tr, Tr], np.repeat(g["text"], 2, 0), seed, [sid, sid + 1], max_generate_length=10, top_k=50)
    assert np.array_equal(codes, g["nrs2"]), (codes, g["nrs2"])
    codes = G.generate(weights, g["refer"], [Tr], g["text"], seed, [sid], max_generate_length=10, top_k=50, input_tokens=g["input_tokens"])
    assert np.array_equal(codes, g["input_tokens_codes"]), (codes, g["input_tokens_codes"])
    codes = G.generate(weights, g["refer"], [Tr], g["text"], seed, [sid], max_generate_length=10, top_k=50, typical_mass=0.9)
    assert np.array_equal(codes, g["typical"]), (codes, g["typical"])
    # input_tokens [2, k] with num_return_sequences = 2: the reference tiles the prefixes to n rows and HF expands each n times -> n * n
    # rows of the ONE prompt, row r starting with input_tokens[(r // n) % 2] and drawing from stream sample_id + r (gpt/model.py:533-537)
    it2 = g["input_tokens2"]
    rows = np.stack([it2[(r // 2) % 2] for r in range(4)])
    codes = G.generate(weights, np.repeat(g["refer"], 4, 0), [Tr] * 4, np.repeat(g["text"], 4, 0), seed, [sid + r for r in range(4)],
                       max_generate_length=10, top_k=50, input_tokens=rows)
    assert np.array_equal(codes, g["input_tokens_nrs2_codes"]), (codes, g["input_tokens_nrs2_codes"])


def test_diffusion_conditioning(weights, golden):
    g = golden("diff_cond")
    cond = D.get_conditioning(weights, g["refer"])
    assert maxabs(cond, g["cond_latent"]) < 1e-5
    ce = D.timestep_independent(weights, g["latent"], g["cond_latent"], 48)
    assert maxabs(ce, g["code_emb"]) < 2e-5


def test_diffusion_forward(weights, golden):
    g = golden("diff_forward")
    oc = D.diffusion_forward(weights, g["x"], g["ts"], g["code_emb"])
    ou = D.diffusion_forward(weights, g["x"], g["ts"], conditioning_free=True)
    assert maxabs(oc, g["out_cond"]) < 5e-5 and maxabs(ou, g["out_uncond"]) < 5e-5


def test_diffusion_sampler_steps(weights, golden):
    g = golden("diff_sampler_steps")
    sched = D.make_schedule()
    seed, sid, T = int(g["seed"]), int(g["sample_id"]), 48
    x = philox.normal(seed, sid, philox.STAGE_DIFF_INIT, 0, 128 * T).reshape(1, 128, T)
    assert maxabs(x, g["x_init"]) < 1e-6
    tr = []
    D.p_sample_loop(weights, sched, g["code_emb"], g["x_init"],
                    lambda i: philox.normal(seed, sid, philox.STAGE_DIFF_STEP, i, 128 * T).reshape(1, 128, T), n_steps=3, trace=tr)
    for rec in tr:
        # teacher-forced per step: eps errors are amplified 153x before the clamp at i=49
        xo, _ = D.p_sample_update(sched, rec["i"], g["x_init"] if rec["i"] == 49 else g[f"x_after_{rec['i'] + 1}"],
                                  D.diffusion_forward(weights, g["x_init"] if rec["i"] == 49 else g[f"x_after_{rec['i'] + 1}"],
                                                      [sched["timestep_map"][rec["i"]]], g["code_emb"]),
                                  D.diffusion_forward(weights, g["x_init"] if rec["i"] == 49 else g[f"x_after_{rec['i'] + 1}"],
                                                      [sched["timestep_map"][rec["i"]]], conditioning_free=True),
                                  philox.normal(seed, sid, philox.STAGE_DIFF_STEP, rec["i"], 128 * T).reshape(1, 128, T))
        assert maxabs(xo, g[f"x_after_{rec['i']}"]) < 2e-3
    # last step (i == 0): noise masked out
    x0 = g["x_before_0"]
    out, _ = D.p_sample_update(sched, 0, x0, D.diffusion_forward(weights, x0, [0], g["code_emb"]),
                               D.diffusion_forward(weights, x0, [0], conditioning_free=True),
                               philox.normal(seed, sid, philox.STAGE_DIFF_STEP, 0, 128 * T).reshape(1, 128, T))
    assert maxabs(out, g["x_after_0"]) < 1e-4


def test_vocoder_stage(weights, golden):
    g = golden("vocoder")
    mel, T = g["mel"], g["mel"].shape[2]
    gg = G.mel_style_encoder(weights, "ref_enc", mel, [T])
    assert maxabs(gg, g["g"]) < 1e-5
    from oracle import ops
    x = ops.conv1d(mel, weights["in_proj.weight"], weights["in_proj.bias"], padding=1)
    _, m_p, logs_p = V.spec_encoder(weights, x, [T])
    assert maxabs(m_p, g["m_p"]) < 2e-5 and maxabs(logs_p, g["logs_p"]) < 2e-5
    mf = np.ones((1, 1, T), np.float32)
    z = V.flow_reverse(weights, g["z_p"], mf, g["g"])
    assert maxabs(z, g["z"]) < 2e-5
    wav = V.generator(weights, g["z"], g["g"])
    assert maxabs(wav, g["wav"]) < 1e-5
    tr = {}
    wav2 = V.infer_flowvae(weights, mel, [T], int(g["seed"]), [int(g["sample_id"])], trace=tr)
    assert maxabs(tr["z_p"], g["z_p"]) < 2e-5
    rms = float(np.sqrt(np.mean((wav2 - g["wav"]) ** 2)))
    assert rms < 1e-4, rms


def test_end_to_end_forced_codes(weights, golden):
    g = golden("e2e_forced")
    wav = pipeline.infer_one(weights, g["text"][0], g["refer"][0], int(g["seed"]), int(g["sample_id"]), forced_codes=g["codes"][0])
    ref = g["wav"][0, 0]
    assert wav.shape == ref.shape
    rms = float(np.sqrt(np.mean((wav - ref) ** 2)))
    assert rms < 1e-3, rms          # north_star tolerance on the 24 kHz waveform


def test_vq_decode_path(weights, golden):
    """infer_gpt's decode (vqvae/model_24k.py:828-845): quantizer.decode -> + vq_ref_enc -> vq_dec -> infer_flowvae."""
    from oracle import vq
    g = golden("vq_path")
    assert maxabs(vq.quantizer_decode(weights, g["codes"]), g["latent"]) < 1e-5
    T = g["refer"].shape[2]
    assert maxabs(G.mel_style_encoder(weights, "vq_ref_enc", g["refer"], [T]), g["g_vq"]) < 1e-4
    mel = vq.vq_decode_mel(weights, g["codes"], g["refer"], [T])
    assert mel.shape == g["recon"].shape
    assert maxabs(mel, g["recon"]) < 1e-4
    wav = vq.infer_gpt_from_codes(weights, g["codes"][0], g["refer"][0], int(g["seed"]), int(g["sample_id"]))
    ref = g["wav"][0, 0]
    assert wav.shape == ref.shape
    assert float(np.sqrt(np.mean((wav - ref) ** 2))) < 1e-3


def test_vq_decode_path_with_an_empty_code_sequence(weights, golden):
    """infer_gpt when the stop token comes first (vqvae/model_24k.py:833-834): a zero latent of 16 frames -> 64 mel frames -> wav,
    against the reference's own infer_gpt run (tests/golden/make_golden_r6.py)."""
    from oracle import vq
    g = golden("infer_gpt_empty")
    T = g["refer"].shape[2]
    mel = vq.vq_decode_mel(weights, np.zeros((1, 0), np.int64), g["refer"], [T])
    assert mel.shape == g["recon"].shape == (1, 128, 64)
    assert maxabs(mel, g["recon"]) < 1e-4
    wav = vq.infer_gpt_from_codes(weights, np.zeros((0,), np.int64), g["refer"][0], int(g["seed"]), int(g["sample_id"]))
    ref = g["wav"][0, 0]
    assert wav.shape == ref.shape == (16384,)
    assert float(np.sqrt(np.mean((wav - ref) ** 2))) < 1e-3


def test_frontend_mel_against_reference_function(golden):
    """oracle.frontend.mel_spectrogram vs the reference's own mel_spectrogram_torch / spectrogram_torch (fixture `frontend`)."""
    from oracle import frontend as FE
    g = golden("frontend")
    mel = FE.mel_spectrogram(g["wav"])
    assert mel.shape == g["mel"].shape
    assert maxabs(mel, g["mel"]) < 1e-4
    # linear magnitude through the same filterbank == exp(log-mel) where the clamp is inactive
    mb = FE.mel_filterbank(24000, 1024, 128)
    lin = np.einsum("mf,bft->bmt", mb, g["spec"])
    ok = lin > 1e-4
    assert maxabs(np.log(lin[ok]), g["mel"][ok]) < 1e-4


def test_frontend_filterbank_and_resampler_known_answers():
    """librosa / torchaudio are absent: known answers of their published algorithms."""
    from oracle import frontend as FE
    # Slaney scale anchors (librosa.hz_to_mel docs): 1000 Hz -> 15 mel, 6400 Hz -> 42 mel; linear below 1 kHz at 200/3 Hz per mel
    assert abs(float(FE._hz_to_mel_slaney(1000.0)) - 15.0) < 1e-9 and abs(float(FE._hz_to_mel_slaney(6400.0)) - 42.0) < 1e-9
    assert abs(float(FE._hz_to_mel_slaney(500.0)) - 7.5) < 1e-9
    mb = FE.mel_filterbank(24000, 1024, 128)
    assert mb.shape == (128, 513) and mb.dtype == np.float32 and (mb >= 0).all()
    # Slaney norm: each triangle integrates to 1 over frequency (bin width 24000/1024 Hz) up to discretisation
    area = mb.sum(1) * (24000 / 1024)
    assert np.all(np.abs(area[4:] - 1.0) < 0.12), (area.min(), area.max())
    peaks = mb.argmax(1)
    assert np.all(np.diff(peaks) >= 0) and peaks[0] >= 1 and peaks[-1] <= 511
    # resampler: output length, unity DC gain away from the edges, a 440 Hz tone survives 44.1 -> 24 kHz with the right phase
    k, width, orig, new = FE.resample_kernel(44100, 24000)
    assert (orig, new, width) == (147, 80, 12) and k.shape == (80, 171)
    x = np.ones((1, 44100), np.float32)
    y = FE.resample(x, 44100, 24000)
    assert y.shape == (1, 24000) and np.abs(y[0, 100:-100] - 1.0).max() < 2e-3
    t = np.arange(44100) / 44100.0
    y = FE.resample(np.sin(2 * np.pi * 440 * t)[None].astype(np.float32), 44100, 24000)[0]
    ref = np.sin(2 * np.pi * 440 * np.arange(24000) / 24000.0)
    assert np.abs(y[200:-200] - ref[200:-200]).max() < 1e-3
    assert FE.resample(x[:, :1000], 24000, 24000).shape == (1, 1000)


def test_vq_encode_path(weights, golden):
    """SynthesizerTrn.encode (vqvae/model_24k.py:877-880): vq_enc + nearest codebook entry, vs the reference fixture."""
    from oracle import vq
    g = golden("vq_encode")
    codes, x_vq = vq.encode(weights, g["mel"])
    assert maxabs(x_vq, g["x_vq"]) < 1e-5
    assert np.array_equal(codes, g["codes"])


def test_torch_backend_of_the_oracle_meets_the_same_fixtures(weights, golden):
    """oracle/ops.py::use_torch - the multi-threaded fp32 torch-CPU building blocks bench.py's cpu_baseline leg times - run through the
    same oracle code as the numpy forms: GPT latents, HF sampling loop, one DiffusionTts.forward, flow-VAE + HiFiGAN against the
    REFERENCE's fixtures.  What is timed as the CPU baseline is therefore a checked restatement too."""
    from oracle import diffusion as D, gpt as G, ops, vocoder as V
    prev = ops.use_torch(True)
    try:
        g = golden("gpt_forced")
        lat = G.latents_teacher_forced(weights, g["refer"], [g["refer"].shape[2]], g["text"].astype(np.int64), g["codes"])
        assert np.abs(lat - g["latent"]).max() < 2e-4
        g = golden("gpt_generate")
        codes = G.generate(weights, g["refer"], [g["refer"].shape[2]], g["text"].astype(np.int64), int(g["seed"]), [int(g["sample_id"])],
                           max_generate_length=10)
        assert np.array_equal(codes, g["codes"])
        g = golden("diff_forward")
        out = D.diffusion_forward(weights, g["x"], g["ts"], g["code_emb"])
        assert np.abs(out - g["out_cond"]).max() < 5e-5
        g = golden("vocoder")
        wav = V.infer_flowvae(weights, g["mel"], [g["mel"].shape[2]], int(g["seed"]), [int(g["sample_id"])])
        assert np.sqrt(np.mean((wav - g["wav"]) ** 2)) < 1e-4
    finally:
        ops.use_torch(prev)
