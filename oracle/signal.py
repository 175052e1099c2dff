"""Oracle: the torchaudio signal ops of the training loop, CPU torch.

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.  **Parity unpinned**: torchaudio is not installed here and the
reference holds no fixture for these ops, so the functions below restate torchaudio 2.x's documented
behaviour (functional.preemphasis, functional.lowpass_biquad -> biquad -> lfilter(clamp=True),
transforms.MelSpectrogram = Spectrogram(hann, center, reflect, power 2, normalized="window") + MelScale(htk,
norm=None)) as the reference calls them:

* ``training.py:151-156``  MelSpectrogram(sample_rate, n_fft=max(w, 512), win_length=w, hop_length=w // 4,
                           n_mels=64, normalized=True) for w in 2**5 .. 2**11
* ``training.py:51-78``    multispectral_reconstruction_loss
* ``training.py:316-318``  lowpass_biquad(x, sample_rate, cutoff_freq)
* ``training.py:333-334``  preemphasis(x, 0.97)
* ``training.py:554``      transforms.Resample(data_sample_rate, sample_rate) (applied per clip by utils.collator)
"""
from __future__ import annotations

import math
from typing import Sequence

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def preemphasis(x: Tensor, coeff: float = 0.97) -> Tensor:
    y = x.clone()
    y[..., 1:] = x[..., 1:] - coeff * x[..., :-1]
    return y


def lowpass_biquad(x: Tensor, sample_rate: int, cutoff_freq: float, q: float = 0.707) -> Tensor:
    w0 = 2 * math.pi * cutoff_freq / sample_rate
    alpha = math.sin(w0) / 2 / q
    b0, b1, b2 = (1 - math.cos(w0)) / 2, 1 - math.cos(w0), (1 - math.cos(w0)) / 2
    a0, a1, a2 = 1 + alpha, -2 * math.cos(w0), 1 - alpha
    b0, b1, b2, a1, a2 = b0 / a0, b1 / a0, b2 / a0, a1 / a0, a2 / a0
    xs = x.reshape(-1, x.shape[-1])
    xs = xs if xs.dtype == torch.float64 else xs.to(torch.float32)   # fp64 only for error-analysis runs of the tests
    y = torch.zeros_like(xs)
    x1 = x2 = y1 = y2 = torch.zeros(xs.shape[0], dtype=xs.dtype)
    for n in range(xs.shape[1]):                      # direct form I, fp32, sequential (the definition)
        xn = xs[:, n]
        yn = b0 * xn + b1 * x1 + b2 * x2 - a1 * y1 - a2 * y2
        y[:, n] = yn
        x2, x1, y2, y1 = x1, xn, y1, yn
    return y.clamp(-1.0, 1.0).reshape(x.shape)        # lfilter(clamp=True)


def hann_periodic(n: int) -> Tensor:
    return torch.hann_window(n, periodic=True, dtype=torch.float64)


def mel_fbanks(n_freqs: int, sample_rate: int, n_mels: int) -> Tensor:
    """torchaudio.functional.melscale_fbanks(n_freqs, 0, sample_rate // 2, n_mels, sample_rate, None, 'htk')
    -> (n_freqs, n_mels) float32."""
    f_max = float(sample_rate // 2)
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min, m_max = 0.0, 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.min(down, up), min=0.0)


def mel_spectrogram(x: Tensor, sample_rate: int, window: int, n_mels: int = 64) -> Tensor:
    """(B, L) -> (B, n_mels, T), T = 1 + L // (window // 4)."""
    n_fft, hop = max(window, 512), window // 4
    win = torch.zeros(n_fft, dtype=torch.float64)
    left = (n_fft - window) // 2
    win[left:left + window] = hann_periodic(window)
    xp = F.pad(x.unsqueeze(1), (n_fft // 2, n_fft // 2), mode="reflect").squeeze(1)
    frames = xp.unfold(-1, n_fft, hop) * win.to(x.dtype)
    spec = torch.fft.rfft(frames, dim=-1)                                  # (B, T, F)
    spec = spec / float(win.pow(2).sum().sqrt())                           # normalized=True -> "window"
    power = spec.real ** 2 + spec.imag ** 2
    return torch.matmul(power, mel_fbanks(n_fft // 2 + 1, sample_rate, n_mels).to(power.dtype)).transpose(1, 2)


def multispectral_reconstruction_loss(original: Tensor, reconstruction: Tensor, sample_rate: int,
                                      windows: Sequence[int] = tuple(2 ** i for i in range(5, 12)),
                                      eps: float = 1e-8, spec_loss_weight: float = 1.0, use_log_l2: bool = True,
                                      scale_alpha: bool = True) -> Tensor:
    """training.py:51-78 on (B, L) signals."""
    loss = 0
    for w in windows:
        alpha = math.sqrt(w / 2) if scale_alpha else 1.0
        so = torch.nan_to_num(mel_spectrogram(original, sample_rate, w))
        sr = torch.nan_to_num(mel_spectrogram(reconstruction, sample_rate, w))
        loss = loss + F.l1_loss(so, sr)
        if use_log_l2:
            loss = loss + alpha * F.mse_loss((so + eps).log(), (sr + eps).log())
        else:
            loss = loss + alpha * F.mse_loss(so, sr)
    return spec_loss_weight * loss


def resample_kernel(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """torchaudio.functional.resample's windowed-sinc table (sinc_interp_hann), entry by entry in float64:
    table[p][k] = scale * sinc(pi t) * cos^2(pi t / (2 lpw)),  t = clamp(base * ((k - width) / of - p / nf), +-lpw),
    base = min(of, nf) * rolloff, scale = base / of, width = ceil(lpw * of / base), K = 2 width + of.
    Returns (table float32 (nf, K), width, of, nf)."""
    g = math.gcd(int(orig_freq), int(new_freq))
    of, nf = int(orig_freq) // g, int(new_freq) // g
    base = min(of, nf) * rolloff
    width = math.ceil(lowpass_filter_width * of / base)
    K = 2 * width + of
    table = torch.zeros(nf, K, dtype=torch.float64)
    for p in range(nf):
        for k in range(K):
            t = base * ((k - width) / of - p / nf)
            t = max(-lowpass_filter_width, min(lowpass_filter_width, t))
            w = math.cos(t * math.pi / lowpass_filter_width / 2) ** 2
            table[p, k] = (1.0 if t == 0 else math.sin(math.pi * t) / (math.pi * t)) * w * base / of
    return table.to(torch.float32), width, of, nf


def resample(x: Tensor, orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99) -> Tensor:
    """transforms.Resample(orig_freq, new_freq)(x): pad (width, width + of), conv1d with the (nf, 1, K) table at
    stride of, interleave the nf phases, crop to ceil(nf * length / of)."""
    if orig_freq == new_freq:
        return x
    table, width, of, nf = resample_kernel(orig_freq, new_freq, lowpass_filter_width, rolloff)
    shape, length = x.shape, x.shape[-1]
    xs = F.pad(x.reshape(-1, length).to(torch.float32), (width, width + of))
    y = F.conv1d(xs[:, None], table[:, None], stride=of)                 # (rows, nf, n)
    y = y.transpose(1, 2).reshape(xs.shape[0], -1)
    target = -(-nf * length // of)
    return y[:, :target].reshape(*shape[:-1], target)
