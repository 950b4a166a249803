"""The oracle against the REFERENCE at the headline sizes (tests/golden/fullsize.npz: subsampled reference outputs at T = 936,
234 codes, T = 5624; make_golden_fullsize.py).  CPU only; pins the oracle where the GPU parity tests use it densely."""
import numpy as np
import pytest

from fullsize_inputs import N_CODES, T, inputs, sub
from oracle import diffusion as D, gpt as G, philox, vocoder as V


def maxabs(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))))


@pytest.fixture(scope="module")
def I():
    return inputs()


@pytest.fixture(scope="module")
def F(golden):
    return golden("fullsize")


def check(out, F, key, tol, ch_stride=None):
    s, t = sub(out, F, ch_stride)
    assert max(maxabs(s, F[key + "_s"]), maxabs(t, F[key + "_t"])) < tol, key


def test_forward_and_p_sample_T936(weights, I, F):
    sched = D.make_schedule()
    ts = [sched["timestep_map"][47]]
    oc = D.diffusion_forward(weights, I["x"], ts, I["code_emb"])
    ou = D.diffusion_forward(weights, I["x"], ts, conditioning_free=True)
    check(oc[0], F, "fwd47_cond", 5e-5)
    check(ou[0], F, "fwd47_uncond", 5e-5)
    ts = [sched["timestep_map"][49]]
    oc = D.diffusion_forward(weights, I["x"], ts, I["code_emb"])
    ou = D.diffusion_forward(weights, I["x"], ts, conditioning_free=True)
    z = philox.normal(1234, 2, philox.STAGE_DIFF_STEP, 49, I["x"].size).reshape(I["x"].shape)
    x1, x0 = D.p_sample_update(sched, 49, I["x"], oc, ou, z)
    check(x1[0], F, "ps49_x", 5e-4)
    check(x0[0], F, "ps49_x0", 2e-3)


def test_attention_T936(weights, I, F):
    check(D.attention_block(weights, "diffusion.layers.3.attn", I["xa"], 16)[0], F, "attn", 2e-5)


def test_gpt_latents_234_codes(weights, I, F):
    lat = G.latents_teacher_forced(weights, I["refer"], [T], I["text"], I["codes"])[0]
    assert lat.shape == (N_CODES, 768)
    assert maxabs(lat[::9], F["gpt_lat_s"]) < 5e-5 and maxabs(lat[-8:], F["gpt_lat_t"]) < 5e-5


def test_vocoder_T936(weights, I, F):
    wav = np.asarray(V.infer_flowvae(weights, I["mel"], [T], 1234, [3])).reshape(-1)
    assert wav.shape[0] == 256 * T
    r = float(np.sqrt(np.mean((wav[::97].astype(np.float64) - F["voc_wav_s"]) ** 2)))
    assert r < 2e-5 and float(F["voc_wav_rms"]) > 1000 * r


def test_oracle_free_sampling_234_tokens_vs_reference_hf_loop(weights, golden):
    """oracle/gpt.py::generate (KV cache) vs the reference's own uncached HF sampling loop over 234 tokens at the 936-frame prompt
    (gpt_generate_fullsize.npz): identical codes, or a first divergence on a CDF edge (margin < 1e-6)."""
    from fullsize_inputs import e2e_inputs
    from oracle import gpt as G
    g, I = golden("gpt_generate_fullsize"), e2e_inputs()
    codes = G.generate(weights, I["refer"], [I["refer"].shape[2]], I["text"].astype(np.int64), int(g["seed"]), [int(g["sample_id"])],
                       max_generate_length=g["codes"].shape[1], suppress_eos=True)
    ref = g["codes"][0]
    got = codes[0][: ref.size]
    if not np.array_equal(got, ref):
        k = int(np.nonzero(got != ref)[0][0])
        assert float(g["f64_margins"][k]) < 1e-6, (k, got[k], ref[k], float(g["f64_margins"][k]))


def test_oracle_vocoder_under_the_signal_weight_set_vs_reference(golden):
    """signal_weights.npz: the reference's infer_flowvae under synthetic_state_dict(0, variant="signal") - the weight set whose waveform
    is driven by z, not by the generator's biases.  Pins the oracle's vocoder on a signal where an error would show (relative RMS)."""
    from detail_tts_amd.weights import select_inference_params, synthetic_state_dict
    from fullsize_inputs import signal_small_inputs
    g, I = golden("signal_weights"), signal_small_inputs()
    P = select_inference_params(synthetic_state_dict(0, variant="signal"))
    rms = lambda a: float(np.sqrt(np.mean(np.asarray(a, np.float64) ** 2)))
    assert rms(g["wav"] - g["wav_z0"]) > 0.5 * rms(g["wav"]) > 0.05
    tr = {}
    wav = V.infer_flowvae(P, I["mel"], [I["mel"].shape[2]], int(g["seed"]), [int(g["sample_id"])], trace=tr)
    assert maxabs(tr["m_p"], g["m_p"]) < 5e-5 and maxabs(tr["logs_p"], g["logs_p"]) < 5e-5 and maxabs(tr["z"], g["z"]) < 1e-4
    assert rms(np.asarray(wav) - g["wav"]) < 2e-5 * rms(g["wav"])
    assert rms(np.asarray(V.generator(P, np.zeros_like(tr["z"]), tr["g"])) - g["wav_z0"]) < 2e-5 * rms(g["wav_z0"])
