"""Oracle for the prompt front-end (TEST INFRASTRUCTURE): wav -> 24 kHz -> log-mel [128, T].

Restates api.py:37-45 and vqvae/utils/data_utils.py:105-155 (`mel_spectrogram_torch`).  Two pieces live in third-party
packages that are ABSENT from this image, so their published algorithms are restated here (parity unpinned for them):
  * torchaudio.transforms.Resample (torchaudio 2.x `functional.resample`, method "sinc_interp_hann",
    lowpass_filter_width 6, rolloff 0.99) — api.py:39
  * librosa.filters.mel (librosa 0.10: Slaney mel scale, htk=False, norm="slaney") — data_utils.py:118-120
The STFT / magnitude / log arithmetic is pinned by tests/golden/frontend.npz (the reference's own function, fed this filterbank).
"""
from __future__ import annotations

import math

import numpy as np

F32 = np.float32


# ------------------------------------------------------------------------------------------------ librosa.filters.mel
def _hz_to_mel_slaney(f):
    f = np.asarray(f, np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, math.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mels)


def _mel_to_hz_slaney(m):
    m = np.asarray(m, np.float64)
    f_sp = 200.0 / 3
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, math.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr, n_fft, n_mels, fmin=0.0, fmax=None):
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax, htk=False, norm='slaney') -> [n_mels, n_fft//2+1] float32."""
    fmax = sr / 2.0 if fmax is None else float(fmax)
    fftfreqs = np.linspace(0.0, sr / 2.0, n_fft // 2 + 1)
    mel_f = _mel_to_hz_slaney(np.linspace(_hz_to_mel_slaney(fmin), _hz_to_mel_slaney(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    w = np.maximum(0.0, np.minimum(lower, upper))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]          # Slaney: constant energy per channel
    return w.astype(F32)


# ------------------------------------------------------------------------------------------------ torchaudio resample
def resample_kernel(orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99):
    """-> (kernel [new, 2*width + orig] float32, width, orig, new) with orig/new reduced by their gcd."""
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
    return (k * window * (base / orig)).astype(F32), width, orig, new


def resample(x, orig_freq, new_freq):
    """x [B, L] -> [B, ceil(L*new/orig)]"""
    x = np.asarray(x, F32)
    if int(orig_freq) == int(new_freq):
        return x.copy()
    k, width, orig, new = resample_kernel(orig_freq, new_freq)
    B, L = x.shape
    xp = np.pad(x, ((0, 0), (width, width + orig)))
    nblk = (xp.shape[1] - k.shape[1]) // orig + 1
    cols = np.lib.stride_tricks.as_strided(xp, shape=(B, nblk, k.shape[1]), strides=(xp.strides[0], xp.strides[1] * orig, xp.strides[1]),
                                           writeable=False)
    y = np.einsum("bnk,pk->bnp", cols, k, optimize=True).reshape(B, -1)
    return np.ascontiguousarray(y[:, : math.ceil(new * L / orig)], F32)


# ------------------------------------------------------------------------------------------------ mel_spectrogram_torch
def mel_spectrogram(y, n_fft=1024, num_mels=128, sampling_rate=24000, hop_size=256, win_size=1024, fmin=0.0, fmax=None, mel_basis=None):
    """y [B, L] in [-1, 1] -> log-mel [B, num_mels, L // hop] (center=False, reflect pad (n_fft - hop)/2 on both sides)."""
    y = np.asarray(y, F32)
    pad = (n_fft - hop_size) // 2
    yp = np.pad(y, ((0, 0), (pad, pad)), mode="reflect")
    nfr = (yp.shape[1] - n_fft) // hop_size + 1
    n = np.arange(win_size, dtype=np.float64)
    win = (0.5 - 0.5 * np.cos(2.0 * math.pi * n / win_size)).astype(F32)       # torch.hann_window (periodic)
    frames = np.lib.stride_tricks.as_strided(yp, shape=(y.shape[0], nfr, n_fft), strides=(yp.strides[0], yp.strides[1] * hop_size, yp.strides[1]),
                                             writeable=False)
    spec = np.fft.rfft(frames.astype(np.float64) * win.astype(np.float64), axis=-1)                 # [B, nfr, n_fft/2+1]
    mag = np.sqrt(spec.real.astype(F32) ** 2 + spec.imag.astype(F32) ** 2 + F32(1e-6)).astype(F32)
    mb = mel_filterbank(sampling_rate, n_fft, num_mels, fmin, fmax) if mel_basis is None else np.asarray(mel_basis, F32)
    mel = np.einsum("mf,btf->bmt", mb, mag).astype(F32)
    return np.log(np.maximum(mel, F32(1e-5))).astype(F32)
