"""Mirror of the reference's prompt front-end helpers (vqvae/utils/data_utils.py:56-155, api.py:37-45) on the MI355X.

    from detail_tts_amd.vqvae.utils.data_utils import mel_spectrogram_torch, HParams, Resample
    audio = Resample(sr, 24000)(audio)                                   # api.py:39 (torchaudio.transforms.Resample)
    spec = mel_spectrogram_torch(audio, hps.data.filter_length, hps.data.n_mel_channels, hps.data.sampling_rate,
                                 hps.data.hop_length, hps.data.win_length, hps.data.mel_fmin, hps.data.mel_fmax)

Both run as HIP kernels behind `dtts_resample` / `dtts_mel_spectrogram` (include/detail_hip.h).  The module-level functions
of the reference have no model argument, so a small runtime holding only the front-end matrices is created on first use
(or pass `rt=model.rt`).
"""
from __future__ import annotations

import torch

from ...config import HParams, load_config          # noqa: F401  (re-export, as in the reference module)

_default_rt = {}


def _runtime(device, rt=None):
    if rt is not None:
        return rt
    device = torch.device(device if torch.device(device).type == "cuda" else "cuda:0")
    key = str(device)
    if key not in _default_rt:
        from ...runtime import Runtime
        _default_rt[key] = Runtime({}, device=device, parts=("frontend",), folded=True)
    return _default_rt[key]


class Resample:
    """torchaudio.transforms.Resample(orig_freq, new_freq) (sinc_interp_hann, lowpass_filter_width 6, rolloff 0.99)."""

    def __init__(self, orig_freq=16000, new_freq=16000, rt=None):
        self.orig_freq, self.new_freq, self.rt = int(orig_freq), int(new_freq), rt

    def __call__(self, waveform):
        w = torch.as_tensor(waveform)
        shape = w.shape
        rt = _runtime(w.device, self.rt)
        x = w.reshape(-1, shape[-1]).to(rt.device, torch.float32).contiguous()
        return rt.resample(x, self.orig_freq, self.new_freq).reshape(*shape[:-1], -1)


def mel_spectrogram_torch(y, n_fft, num_mels, sampling_rate, hop_size, win_size, fmin, fmax, center=False, *, lengths=None, rt=None):
    """vqvae/utils/data_utils.py:105-155.  y [B, L] (or [L]) in [-1, 1] -> log-mel [B, num_mels, L // hop_size] on the GPU."""
    if center:
        raise NotImplementedError("center=True is never used by the reference")
    y = torch.as_tensor(y)
    if y.dim() == 1:
        y = y[None]
    rt = _runtime(y.device, rt)
    d = rt.cfg["data"]
    want = (d["filter_length"], d["n_mel_channels"], d["sampling_rate"], d["hop_length"], d["win_length"], float(d.get("mel_fmin", 0.0)), d.get("mel_fmax"))
    got = (n_fft, num_mels, sampling_rate, hop_size, win_size, float(fmin), fmax)
    if want != got:
        raise ValueError(f"front-end matrices were packed for {want}, called with {got}")
    return rt.mel_spectrogram(y.to(rt.device, torch.float32).contiguous(), lengths)


def spectrogram_torch(y, n_fft, sampling_rate, hop_size, win_size, center=False, *, lengths=None, rt=None):
    """vqvae/utils/data_utils.py:56-87 (imported by api.py:29).  y [B, L] -> linear magnitude spectrogram [B, n_fft // 2 + 1, L // hop_size]."""
    if center:
        raise NotImplementedError("center=True is never used by the reference")
    y = torch.as_tensor(y)
    if y.dim() == 1:
        y = y[None]
    rt = _runtime(y.device, rt)
    d = rt.cfg["data"]
    want = (d["filter_length"], d["sampling_rate"], d["hop_length"], d["win_length"])
    if want != (n_fft, sampling_rate, hop_size, win_size):
        raise ValueError(f"front-end matrices were packed for {want}, called with {(n_fft, sampling_rate, hop_size, win_size)}")
    return rt.spectrogram(y.to(rt.device, torch.float32).contiguous(), lengths)
