"""Waveform comparison metrics -- TEST INFRASTRUCTURE (see mvts_oracle.py header for who may import this).

The reference has no evaluation metric on its inference path; the mel analysis below uses the reference's data
settings (configs/config.json: sampling_rate 44100, filter_length 2048, hop_length 512, win_length 2048,
n_mel_channels 80, mel_fmin 0, mel_fmax null) with a build-owned HTK-style triangular filterbank, because the
reference's mel basis comes from librosa (mel_processing.py:64-95), which is not installed here.
"""
from __future__ import annotations

import math

import numpy as np
import torch

SR, N_FFT, HOP, WIN, N_MELS = 44100, 2048, 512, 2048, 80


def _hz_to_mel(f):
    return 2595.0 * np.log10(1.0 + np.asarray(f, dtype=np.float64) / 700.0)


def _mel_to_hz(m):
    return 700.0 * (10.0 ** (np.asarray(m, dtype=np.float64) / 2595.0) - 1.0)


def mel_filterbank(sr: int = SR, n_fft: int = N_FFT, n_mels: int = N_MELS, fmin: float = 0.0, fmax: float = None):
    fmax = sr / 2.0 if fmax is None else fmax
    freqs = np.linspace(0.0, sr / 2.0, n_fft // 2 + 1)
    pts = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fb = np.zeros((n_mels, n_fft // 2 + 1), dtype=np.float64)
    for i in range(n_mels):
        lo, ce, hi = pts[i], pts[i + 1], pts[i + 2]
        up = (freqs - lo) / max(ce - lo, 1e-9)
        dn = (hi - freqs) / max(hi - ce, 1e-9)
        fb[i] = np.maximum(0.0, np.minimum(up, dn))
    return torch.from_numpy(fb).to(torch.float32)


_FB = None


def log_mel(wav: torch.Tensor) -> torch.Tensor:
    """wav [..., N] float32 -> log-mel [..., 80, frames] with log(clamp(x, 1e-5)) compression."""
    global _FB
    if _FB is None:
        _FB = mel_filterbank()
    w = wav.reshape(-1, wav.shape[-1]).to(torch.float32).cpu()
    pad = (N_FFT - HOP) // 2
    w = torch.nn.functional.pad(w[:, None, :], (pad, pad), mode="reflect")[:, 0]
    spec = torch.stft(w, N_FFT, hop_length=HOP, win_length=WIN, window=torch.hann_window(WIN), center=False,
                      return_complex=True).abs()
    mel = torch.matmul(_FB, spec)
    return torch.log(torch.clamp(mel, min=1e-5))


def log_mel_l1(a: torch.Tensor, b: torch.Tensor) -> float:
    """Mean absolute difference of the two log-mel spectrograms."""
    return float((log_mel(a) - log_mel(b)).abs().mean())


def max_abs(a: torch.Tensor, b: torch.Tensor) -> float:
    return float((a.detach().cpu().to(torch.float64) - b.detach().cpu().to(torch.float64)).abs().max())
