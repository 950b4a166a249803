"""Prompt front-end constants: the DFT and mel matrices of `mel_spectrogram_torch`
(vqvae/utils/data_utils.py:105-155) and the polyphase kernel of torchaudio's sinc resampler (api.py:39), built on the host in
float64 and packed with the other weights.  torchaudio and librosa are not dependencies: their published algorithms
(torchaudio 2.x `functional.resample` "sinc_interp_hann" width 6 rolloff 0.99; librosa 0.10 `filters.mel` Slaney scale + norm)
are restated here.
"""
from __future__ import annotations

import math

import numpy as np

F32 = np.float32


def _hz_to_mel(f):
    f = np.asarray(f, np.float64)
    f_sp, min_log_hz, logstep = 200.0 / 3, 1000.0, math.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_hz / f_sp + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, f / f_sp)


def _mel_to_hz(m):
    m = np.asarray(m, np.float64)
    f_sp, min_log_hz, logstep = 200.0 / 3, 1000.0, math.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr, n_fft, n_mels, fmin=0.0, fmax=None):
    """[n_mels, n_fft // 2 + 1] triangular filters on the Slaney mel scale, each normalised to constant energy per channel."""
    fmax = sr / 2.0 if fmax is None else float(fmax)
    freqs = np.linspace(0.0, sr / 2.0, n_fft // 2 + 1)
    edges = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    d = np.diff(edges)
    ramps = edges[:, None] - freqs[None, :]
    w = np.maximum(0.0, np.minimum(-ramps[:-2] / d[:-1, None], ramps[2:] / d[1:, None]))
    w *= (2.0 / (edges[2:n_mels + 2] - edges[:n_mels]))[:, None]
    return w.astype(F32)


def dft_matrix(n_fft, win_size):
    """[n_fft + 2, n_fft]: rows 0..n_fft/2 = hann[k] cos(2 pi f k / N), rows n_fft/2+1.. = -hann[k] sin(...) (periodic Hann)."""
    assert win_size == n_fft, "win_size != n_fft is not used by the reference configs"
    k = np.arange(n_fft, dtype=np.float64)
    win = 0.5 - 0.5 * np.cos(2.0 * math.pi * k / win_size)
    f = np.arange(n_fft // 2 + 1, dtype=np.float64)[:, None]
    ang = 2.0 * math.pi * f * k[None, :] / n_fft
    return np.concatenate([np.cos(ang) * win, -np.sin(ang) * win], 0).astype(F32)


def resample_kernel(orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99):
    """-> (kernel [new, 2*width + orig] float32, width, orig, new), frequencies reduced by their gcd."""
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    idx = np.arange(-width, width + orig, dtype=np.float64)[None, :] / orig
    t = (np.arange(0, -new, -1, dtype=np.float64)[:, None] / new + idx) * base
    t = np.clip(t, -lowpass_filter_width, lowpass_filter_width)
    window = np.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    with np.errstate(divide="ignore", invalid="ignore"):
        k = np.where(t == 0, 1.0, np.sin(t) / t)
    return np.ascontiguousarray(k * window * (base / orig), F32), width, orig, new


def pack_frontend(pk, cfg):
    d = cfg["data"]
    n_fft, win, n_mels, sr = d["filter_length"], d["win_length"], d["n_mel_channels"], d["sampling_rate"]
    pk.conv("frontend.dft", dft_matrix(n_fft, win)[:, :, None])
    pk.conv("frontend.mel", mel_filterbank(sr, n_fft, n_mels, d.get("mel_fmin", 0.0), d.get("mel_fmax"))[:, :, None])
