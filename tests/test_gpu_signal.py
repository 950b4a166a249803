"""GPU parity gates that BITE.

With the plain seed-0 synthetic weights the HiFiGAN waveform is 99 % bias, so a waveform comparison cannot see an error upstream of the
generator.  `synthetic_state_dict(0, variant="signal")` rescales the vocoder so that 6/7 of the waveform is driven by z; the fixtures
used here were produced by the REFERENCE ITSELF under that weight set (tests/golden/make_golden_r4.py):

  * signal_weights.npz        infer_flowvae on a small mel (g, m_p, logs_p, z, waveform),
  * e2e_fullsize_signal.npz   the reference's own SynthesizerTrn.infer at the headline configuration (234 codes, 50 steps),
  * longform.npz              configs[4] (60 s, T = 5624): DiffusionTts.forward cond + uncond (seed-0 weights) and infer_flowvae under the
                              signal weights, against which the ONE-SHOT and the STREAMED vocoder are both compared.

The last tests inject a 1 % error into one p_sample / into the mel and assert that the gates above would have failed."""
import numpy as np
import pytest

from conftest import tol
from fullsize_inputs import N_CODES, T, T_LONG, e2e_inputs, longform_inputs, signal_small_inputs, sub

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

REL_WAV = 6e-5          # relative waveform RMS gate under the signal weights: ~20 x the measured 1.7e-6 .. 3.4e-6 (profiles/r04_measured_errors.txt)


def maxabs(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))))


def rms(a, b=0.0):
    return float(np.sqrt(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2)))


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()


def host(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def weights_signal():
    from detail_tts_amd.weights import select_inference_params, synthetic_state_dict
    return select_inference_params(synthetic_state_dict(0, variant="signal"))


@pytest.fixture(scope="module")
def synth(weights_signal):
    from detail_tts_amd.vqvae.model_24k import SynthesizerTrn
    return SynthesizerTrn(weights_signal, folded=True)


def test_infer_flowvae_signal_weights_vs_reference(synth, golden):
    """infer_flowvae (vqvae/model_24k.py:848-863) under the signal weights: every intermediate and the waveform vs the reference."""
    g, I = golden("signal_weights"), signal_small_inputs()
    rt = synth.rt
    assert rms(g["wav"], g["wav_z0"]) > 0.5 * rms(g["wav"]) > 0.05          # the weight set does what it is for
    gg = host(rt.mel_style("ref_enc", dev(I["mel"])))
    tol("signal_small_g_maxabs", maxabs(gg.reshape(-1), g["g"].reshape(-1)), 5e-6)
    m_p, logs_p = rt.op_enc_p(dev(I["mel"]))
    tol("signal_small_m_p_maxabs", maxabs(host(m_p), g["m_p"]), 2e-5 * max(1.0, float(np.abs(g["m_p"]).max())))
    tol("signal_small_logs_p_maxabs", maxabs(host(logs_p), g["logs_p"]), 2e-5)
    wav, z = rt.vocoder(dev(I["mel"]), int(g["seed"]), [int(g["sample_id"])], return_z=True)
    tol("signal_small_z_maxabs", maxabs(host(z), g["z"]), 2e-5 * max(1.0, float(np.abs(g["z"]).max())))
    tol("signal_small_wav_rel_rms", rms(host(wav), g["wav"]) / rms(g["wav"]), REL_WAV)
    w0 = host(rt.generator(torch.zeros_like(z), dev(g["g"][:, :, 0])))
    tol("signal_small_wav_z0_rel_rms", rms(w0, g["wav_z0"]) / rms(g["wav_z0"]), REL_WAV)


@pytest.mark.parametrize("x3", [1, 0])
def test_e2e_234_codes_signal_weights_vs_reference_waveform(synth, golden, x3):
    """north_star's waveform criterion where it can fail: the reference's own SynthesizerTrn.infer (vqvae/model_24k.py:774-810) at the
    headline configuration under the signal weights - relative waveform RMS, both kernel sets, B = 1."""
    E, EI = golden("e2e_fullsize_signal"), e2e_inputs()
    synth.rt.set_option("conv_x3", x3)
    try:
        wav = synth.infer(torch.from_numpy(EI["text"]), torch.tensor([61]), torch.from_numpy(EI["refer"]), torch.tensor([T]),
                          seed=int(E["seed"]), sample_ids=[int(E["sample_id"])], forced_codes=[EI["codes"][0]])
    finally:
        synth.rt.set_option("conv_x3", 1)
    w = host(wav)[0, 0]
    assert w.shape == E["wav"].shape and float(E["wav_rms"]) > 0.1
    r = rms(w, E["wav"])
    print(f"\n[e2e 234 codes, signal weights, conv_x3={x3}] waveform RMS error {r:.3e} on a {float(E['wav_rms']):.3f} RMS signal")
    tol(f"e2e_signal_wav_rel_rms_x3={x3}", r / float(E["wav_rms"]), REL_WAV)
    assert r < 1e-3                                   # north_star's absolute figure, now on an O(0.2) signal


def test_e2e_signal_weights_row_of_ragged_B8(synth, golden):
    """The same utterance as row 5 of a ragged batch of 8 (batch-8 kernel shapes in all three stages)."""
    from test_gpu_fullsize import _ragged_batch8
    E, EI = golden("e2e_fullsize_signal"), e2e_inputs()
    refer, rl, text, tl, codes, n = _ragged_batch8(EI)
    wav, lens = synth.infer(torch.from_numpy(text), torch.tensor(tl), torch.from_numpy(refer), torch.tensor(rl), batch=True,
                            seed=int(E["seed"]), sample_ids=[100, 101, 102, 103, 104, int(E["sample_id"]), 106, 107],
                            forced_codes=codes, return_lengths=True)
    w = host(wav)[5, 0, : lens[5]]
    tol("e2e_signal_row5_of_B8_wav_rel_rms", rms(w, E["wav"]) / float(E["wav_rms"]), REL_WAV)


def _chain(rt, code_emb, seed, sid, perturb_after=None, factor=1.01):
    """the 50 p_sample steps one by one through dtts_diff_p_sample (same Philox noise as dtts_diff_sample); optionally x *= factor
    after sampling step `perturb_after`"""
    from oracle import philox
    x = dev(philox.normal(seed, sid, philox.STAGE_DIFF_INIT, 0, 128 * T).reshape(1, 128, T))
    for i in range(49, -1, -1):
        x = rt.diff_p_sample(x, code_emb, i, seed, [sid], lens=[T])
        if perturb_after == i:
            x = x * factor
    return x


def test_a_one_percent_error_in_one_p_sample_or_in_the_mel_fails_the_gates(synth, golden):
    """The gates bite: the clean step-by-step chain meets them; the same chain with ONE sampler state scaled by 1.01 (after step 25),
    or with the mel scaled by 1.01 in front of the vocoder, exceeds the mel and the waveform gate by a wide margin.  (Under the seed-0
    weights the waveform gate of round 3 - 1e-3 absolute - passed with the diffusion output replaced by zeros.)"""
    from detail_tts_amd.vqvae.model_24k import denormalize_torch_mel
    E, EI = golden("e2e_fullsize_signal"), e2e_inputs()
    rt = synth.rt
    seed, sid = int(E["seed"]), int(E["sample_id"])
    refer = dev(EI["refer"])
    lat = rt.gpt_latents(refer, [T], [EI["text"][0]], [EI["codes"][0]])
    code_emb = rt.diff_timestep_independent(lat, rt.diff_conditioning(refer, [T]), [N_CODES])

    def errors(mel):
        s, t = sub(host(mel)[0], E)
        e_mel = max(maxabs(s, E["mel_s"]), maxabs(t, E["mel_t"]))
        w = host(rt.vocoder(mel, seed, [sid]))[0, 0]
        return e_mel, rms(w, E["wav"]) / float(E["wav_rms"])

    mel = denormalize_torch_mel(_chain(rt, code_emb, seed, sid))
    e_mel, e_wav = errors(mel)
    tol("bite_clean_mel_maxabs", e_mel, 2e-3)
    tol("bite_clean_wav_rel_rms", e_wav, REL_WAV)
    b_mel, b_wav = errors(denormalize_torch_mel(_chain(rt, code_emb, seed, sid, perturb_after=25)))
    print(f"\n[1 % error after p_sample 25] mel max-abs {b_mel:.3e} (gate 2e-3), waveform relative RMS {b_wav:.3e} (gate {REL_WAV:.0e})")
    assert b_mel > 10 * 2e-3 and b_wav > 10 * REL_WAV, (b_mel, b_wav)
    c_mel, c_wav = errors(mel * 1.01)
    print(f"[1 % error on the mel] mel max-abs {c_mel:.3e}, waveform relative RMS {c_wav:.3e}")
    assert c_mel > 10 * 2e-3 and c_wav > 10 * REL_WAV, (c_mel, c_wav)
    d_wav = rms(host(rt.vocoder(torch.zeros_like(mel) - 5.0, seed, [sid]))[0, 0], E["wav"]) / float(E["wav_rms"])
    assert d_wav > 0.3, d_wav                          # stage B deleted: the waveform is simply different


# ---------------------------------------------------------------------------------------------------- configs[4]: 60 s, T = 5624
@pytest.fixture(scope="module")
def LF(golden):
    return golden("longform")


def test_diffusion_forward_T5624_vs_reference(weights, LF):
    """DiffusionTts.forward (vqvae/diff_model.py:262-322) at the 60 s length, cond + uncond, against the reference's own output
    (88 key tiles per attention row, 30 N-tiles per conv, the two-pass GroupNorm kernel of rows longer than 1022)."""
    from detail_tts_amd.runtime import Runtime
    rt = Runtime(weights, folded=True, parts=("diffusion",))
    I = longform_inputs()
    for x3 in (1, 0):
        rt.set_option("conv_x3", x3)
        oc = host(rt.diff_forward(dev(I["x"]), 47, dev(I["code_emb"])))[0]
        ou = host(rt.diff_forward(dev(I["x"]), 47, None, cond_free=True))[0]
        for name, o in (("cond", oc), ("uncond", ou)):
            s, t = sub(o, LF)
            e = max(maxabs(s, LF[f"fwd47_{name}_s"]), maxabs(t, LF[f"fwd47_{name}_t"]))
            tol(f"fwd_T5624_{name}_x3={x3}_maxabs", e, 3e-4)
    rt.set_option("conv_x3", 1)


@pytest.mark.parametrize("chunk", [0, 256, 100])
def test_vocoder_60s_one_shot_and_streamed_vs_reference_waveform(synth, LF, chunk):
    """infer_flowvae (vqvae/model_24k.py:848-863) at T = 5624 under the signal weights: the one-shot vocoder AND dtts_vocoder_stream
    (256-frame windows: the fixture keeps the reference's samples densely around every window seam; and 100-frame windows, seams at
    other places) against the REFERENCE's waveform - not against each other."""
    I = longform_inputs()
    rt = synth.rt
    seed, sid = int(LF["seed"]), int(LF["sample_id"])
    if chunk == 0:
        wav, z = rt.vocoder(dev(I["mel"]), seed, [sid], return_z=True)
        s, t = sub(host(z)[0], LF)
        tol("voc_T5624_z_maxabs", max(maxabs(s, LF["voc_z_s"]), maxabs(t, LF["voc_z_t"])), 2e-5 * max(1.0, float(np.abs(LF["voc_z_s"]).max())))
    else:
        wav = rt.vocoder(dev(I["mel"]), seed, [sid], stream_chunk=chunk)
    w = host(wav)[0, 0]
    assert w.shape == (256 * T_LONG,)
    ref_rms = float(LF["voc_wav_rms"])
    assert ref_rms > 0.1
    tol(f"voc_T5624_chunk{chunk}_wav_sub_rel_rms", rms(w[:: int(LF["wav_stride"])], LF["voc_wav_s"]) / ref_rms, REL_WAV)
    tol(f"voc_T5624_chunk{chunk}_wav_tail_rel_rms", rms(w[-2048:], LF["voc_wav_t"]) / ref_rms, REL_WAV)
    h = int(LF["seam_half"])
    worst = max(rms(w[p - h: p + h], ref) for p, ref in zip(LF["seam_pos"], LF["voc_wav_seams"]))
    tol(f"voc_T5624_chunk{chunk}_wav_seams_rel_rms", worst / ref_rms, REL_WAV)
    tol(f"voc_T5624_chunk{chunk}_wav_seams_maxabs", max(maxabs(w[p - h: p + h], ref) for p, ref in zip(LF["seam_pos"], LF["voc_wav_seams"])), 5e-5)
