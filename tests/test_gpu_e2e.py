"""GPU parity: the whole path through the reference-shaped Python surface (load_model / SynthesizerTrn.infer)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def rms(a, b):
    return float(np.sqrt(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2)))


@pytest.fixture(scope="module")
def model(weights):
    from detail_tts_amd.vqvae.model_24k import SynthesizerTrn
    return SynthesizerTrn(weights, folded=True)


def test_e2e_forced_codes_vs_reference_golden(model, golden):
    """SynthesizerTrn.infer with forced codes + Philox noise vs the waveform the REFERENCE produced (golden)."""
    g = golden("e2e_forced")
    text = torch.from_numpy(g["text"])
    wav = model.infer(text, torch.tensor([text.shape[1]]), torch.from_numpy(g["refer"]), torch.tensor([g["refer"].shape[2]]),
                      seed=int(g["seed"]), sample_ids=[int(g["sample_id"])], forced_codes=[g["codes"][0]])
    wav = wav.cpu().numpy()
    assert wav.shape == g["wav"].shape
    r = rms(wav, g["wav"])
    assert r < 1e-3, r                       # north_star: <= 1e-3 RMS on the 24 kHz waveform
    assert float(np.sqrt(np.mean(g["wav"] ** 2))) > 20 * r


def test_e2e_free_sampling_batch_vs_oracle(model, weights):
    """Two utterances of different text / prompt length in one batch == each run alone through the oracle."""
    from oracle import pipeline
    rs = np.random.RandomState(33)
    refer = (rs.randn(2, 128, 44) * 2 - 5).astype(np.float32)
    rl = [44, 30]
    texts = [np.concatenate([rs.randint(3, 255, 8), [0]]), np.concatenate([rs.randint(3, 255, 5), [0]])]
    text = np.zeros((2, 9), np.int32)
    for i, t in enumerate(texts):
        text[i, :len(t)] = t
    wav, lens = model.infer(torch.from_numpy(text), torch.tensor([9, 6]), torch.from_numpy(refer), torch.tensor(rl), batch=True,
                            seed=4321, sample_ids=[70, 71], max_generate_length=5, suppress_eos=True, return_lengths=True)
    wav = wav.cpu().numpy()
    for b in range(2):
        ref = pipeline.infer_one(weights, texts[b], refer[b, :, :rl[b]], 4321, 70 + b, max_generate_length=5, suppress_eos=True)
        assert lens[b] == ref.shape[0] == 4 * 1024
        r = rms(wav[b, 0, :lens[b]], ref)
        assert r < 1e-3, (b, r)


def test_load_model_surface():
    from detail_tts_amd.prepare.load_infer import load_model
    m = load_model("vqvae", "synthetic:0", None, "cuda:0")
    assert hasattr(m, "infer") and hasattr(m, "infer_flowvae") and hasattr(m.gpt, "inference_speech_tortoise")
    assert hasattr(m.diffusion, "get_conditioning") and hasattr(m.dec, "forward")
