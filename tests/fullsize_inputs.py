"""Seeded inputs of tests/golden/fullsize.npz (same recipe as tests/golden/make_golden_fullsize.py::inputs; the reference's
outputs for them are stored subsampled in the fixture)."""
import numpy as np

T = 936
N_CODES = 234
T_LONG = 5624
L_TEXT = 60


def inputs(seed=11):
    rs = np.random.RandomState(seed)
    d = {}
    d["x"] = rs.randn(1, 128, T).astype(np.float32)
    d["code_emb"] = (rs.randn(1, 768, T) * 0.5).astype(np.float32)
    d["xa"] = rs.randn(1, 768, T).astype(np.float32)
    d["xa_long"] = rs.randn(1, 768, T_LONG).astype(np.float32)
    d["mel"] = (rs.randn(1, 128, T) * 2 - 5).astype(np.float32)
    d["refer"] = (rs.randn(1, 128, T) * 2 - 5).astype(np.float32)
    d["text"] = np.concatenate([rs.randint(3, 255, (1, L_TEXT)), [[0]]], 1).astype(np.int32)
    d["codes"] = rs.randint(0, 8192, (1, N_CODES)).astype(np.int64)
    return d


def sub(a, g, ch_stride=None):
    """[C, T] -> the fixture's (strided block, tail block)"""
    cs = int(g["ch_stride"]) if ch_stride is None else ch_stride
    return a[::cs, ::int(g["t_stride"])], a[:, -int(g["tail"]):]


def e2e_inputs(seed=21):
    """Inputs of tests/golden/e2e_fullsize.npz / gpt_generate_fullsize.npz (make_golden_e2e_fullsize.py): the headline
    configuration - 10 s prompt (936 frames), 60 text ids + the appended 0 (api.py:24-25), 234 forced codes."""
    rs = np.random.RandomState(seed)
    d = {"seed_inputs": seed}
    d["refer"] = (rs.randn(1, 128, T) * 2 - 5).astype(np.float32)
    d["text"] = np.concatenate([rs.randint(3, 255, (1, L_TEXT)), [[0]]], 1).astype(np.int32)
    d["codes"] = rs.randint(0, 8192, (1, N_CODES)).astype(np.int64)
    return d


def e2e_inputs_b(seed=22):
    """A second, SHORTER utterance (7.5 s prompt = 700 frames, 40 text ids, 150 codes -> T = 600): with e2e_inputs() it makes a ragged batch
    whose rows are both pinned by the reference's own waveforms (tests/golden/e2e_fullsize_b.npz)."""
    rs = np.random.RandomState(seed)
    d = {"seed_inputs": seed}
    d["refer"] = (rs.randn(1, 128, 700) * 2 - 5).astype(np.float32)
    d["text"] = np.concatenate([rs.randint(3, 255, (1, 40)), [[0]]], 1).astype(np.int32)
    d["codes"] = rs.randint(0, 8192, (1, 150)).astype(np.int64)
    return d


def longform_inputs(seed=31):
    """Inputs of tests/golden/longform.npz (make_golden_r4.py): BASELINE.json configs[4] - one 60 s utterance, T = 5624 mel frames."""
    rs = np.random.RandomState(seed)
    d = {"seed_inputs": seed}
    d["x"] = rs.randn(1, 128, T_LONG).astype(np.float32)
    d["code_emb"] = (rs.randn(1, 768, T_LONG) * 0.5).astype(np.float32)
    d["mel"] = (rs.randn(1, 128, T_LONG) * 2 - 5).astype(np.float32)
    return d


def signal_small_inputs(seed=41, T=48):
    """Inputs of the small vocoder fixture under the "signal" weight variant (tests/golden/signal_weights.npz)."""
    rs = np.random.RandomState(seed)
    return {"seed_inputs": seed, "mel": (rs.randn(1, 128, T) * 2 - 5).astype(np.float32)}


# waveform subsample of the 60 s fixture: every WAV_STRIDE-th sample + dense windows around the seams of the streamed vocoder
WAV_STRIDE = 11
SEAM_FRAMES = 256          # dtts_vocoder_stream chunk_frames of the test
SEAM_HALF = 1024           # samples kept on each side of a seam
