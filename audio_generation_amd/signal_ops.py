"""The training loop's signal ops on HIP kernels -- what ``networks/training.py`` takes from
torchaudio: ``MelSpectrogram`` (:151-156), ``multispectral_reconstruction_loss`` (:51-78),
``functional.lowpass_biquad`` (:316-318), ``functional.preemphasis`` (:333-334) and ``transforms.Resample`` (:554,
applied per clip by ``utils.collator``, utils.py:157-158).

**Parity unpinned**: torchaudio is not installed in the build container and the reference holds no fixture
for these ops; behaviour is restated from torchaudio's documentation (``oracle/signal.py``) and the kernels
are tested against that restatement.

The spectrogram is a framed DFT = a polyphase conv on the MFMA conv kernel (``csrc/spectral.hip``), the
mel projection + power a small VALU kernel, the loss means the reduction kernels of the discriminator loss;
every op has a hand-written backward (the loss is differentiated w.r.t. the reconstruction).
"""
from __future__ import annotations

import ctypes
import math
from typing import Optional, Sequence

import torch
from torch import nn

from . import _lib, ops
from .ops import _f32c, _need_gpu, _ptr, _stream

Tensor = torch.Tensor


def _rows(x: Tensor):
    return x.numel() // x.shape[-1]


class _Preemphasis(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, coeff: float):
        ctx.coeff = coeff
        return _preemph_raw(x.detach(), coeff, 0)

    @staticmethod
    def backward(ctx, g: Tensor):
        return _preemph_raw(g.contiguous(), ctx.coeff, 1), None


def _preemph_raw(x: Tensor, coeff: float, adjoint: int) -> Tensor:
    lib = _lib.load()
    _need_gpu(x)
    x = _f32c(x)
    y = torch.empty_like(x)
    _lib.check(lib.agx_preemphasis(_ptr(x), _ptr(y), _rows(x), x.shape[-1], float(coeff), adjoint, _stream()),
               "agx_preemphasis")
    return y


def preemphasis(waveform: Tensor, coeff: float = 0.97) -> Tensor:
    """torchaudio.functional.preemphasis."""
    return _Preemphasis.apply(waveform, coeff)


def lowpass_biquad(waveform: Tensor, sample_rate: int, cutoff_freq: float, Q: float = 0.707) -> Tensor:
    """torchaudio.functional.lowpass_biquad (no gradient: the reference applies it to the input batch only)."""
    lib = _lib.load()
    _need_gpu(waveform)
    x = _f32c(waveform.detach())
    y = torch.empty_like(x)
    _lib.check(lib.agx_lowpass_biquad(_ptr(x), _ptr(y), _rows(x), x.shape[-1], float(sample_rate), float(cutoff_freq),
                                      float(Q), _stream()), "agx_lowpass_biquad")
    return y


class Resample(nn.Module):
    """torchaudio.transforms.Resample(orig_freq, new_freq) with its defaults (sinc_interp_hann,
    lowpass_filter_width 6, rolloff 0.99): the reference's ``resampler`` (training.py:554).  The windowed-sinc
    table (new_freq / gcd phases x 2 width + orig_freq / gcd taps) is built once in float64 and kept in float32;
    the interpolation runs on ``agx_resample`` (csrc/spectral.hip)."""

    def __init__(self, orig_freq: int = 16000, new_freq: int = 16000, resampling_method: str = "sinc_interp_hann",
                 lowpass_filter_width: int = 6, rolloff: float = 0.99):
        super().__init__()
        if resampling_method != "sinc_interp_hann":
            raise NotImplementedError("Resample: only torchaudio's default sinc_interp_hann window")
        if int(orig_freq) != orig_freq or int(new_freq) != new_freq or orig_freq <= 0 or new_freq <= 0:
            raise ValueError("Resample: frequencies must be positive integers")
        if lowpass_filter_width <= 0:
            raise ValueError("Low pass filter width should be positive.")
        self.orig_freq, self.new_freq = int(orig_freq), int(new_freq)
        g = math.gcd(self.orig_freq, self.new_freq)
        self.of, self.nf = self.orig_freq // g, self.new_freq // g
        base = min(self.of, self.nf) * rolloff
        self.width = math.ceil(lowpass_filter_width * self.of / base)
        taps = torch.arange(-self.width, self.width + self.of, dtype=torch.float64) / self.of          # (K,)
        phase = -torch.arange(self.nf, dtype=torch.float64) / self.nf                                  # (nf,)
        t = ((phase[:, None] + taps[None, :]) * base).clamp(-lowpass_filter_width, lowpass_filter_width)
        window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
        t = t * math.pi
        sinc = torch.where(t == 0, torch.ones_like(t), torch.sin(t) / torch.where(t == 0, torch.ones_like(t), t))
        self.register_buffer("kernel", (sinc * window * (base / self.of)).to(torch.float32), persistent=False)

    def forward(self, waveform: Tensor) -> Tensor:
        if self.orig_freq == self.new_freq:
            return waveform
        lib = _lib.load()
        _need_gpu(waveform, self.kernel)
        x = _f32c(waveform)
        rows, length = _rows(x), x.shape[-1]
        out_len = int(lib.agx_resample_out_len(length, self.of, self.nf))
        y = torch.empty(*x.shape[:-1], out_len, dtype=torch.float32, device=x.device)
        for r0 in range(0, rows, 65535):            # the row index rides on grid.y
            r1 = min(rows, r0 + 65535)
            _lib.check(lib.agx_resample(_ptr(x.view(rows, length)[r0:r1]), _ptr(self.kernel), _ptr(y.view(rows, out_len)[r0:r1]),
                                        r1 - r0, length, self.of, self.nf, self.width, _stream()), "agx_resample")
        return y


def collator(batch, size: int = 72000, resampler: Optional[nn.Module] = None):
    """``utils.collator`` (utils.py:149-175): per clip ``x = item[0]`` (the label is dropped), resample, then zero-pad
    at a random split or crop at a random offset to ``size`` samples.  As in the reference a clip whose length
    already equals ``size`` is left out of the returned list (its ``if / elif`` has no branch for it), and the
    random offsets come from ``torch.randint`` on the default generator.  Clips are moved to the GPU for the
    resampler (``Resample`` has no CPU path) and come back on the device they arrived on."""
    out = []
    for item in batch:
        x = item[0]
        if resampler is not None:
            dev = x.device
            x = resampler(x.cuda() if not x.is_cuda else x).to(dev)
        n = x.shape[-1]
        if n < size:
            diff = size - n
            split = int(torch.randint(0, diff, (1,)).item())
            out.append(torch.cat([x.new_zeros((x.shape[0], split)), x, x.new_zeros((x.shape[0], diff - split))], dim=-1))
        elif n > size:
            start = int(torch.randint(0, n - size, (1,)).item())
            out.append(x[:, start:start + size])
    return out


def melscale_fbanks(n_freqs: int, sample_rate: int, n_mels: int) -> Tensor:
    """torchaudio.functional.melscale_fbanks(n_freqs, 0, sample_rate // 2, n_mels, sample_rate, None, 'htk'):
    a constant of the transform, built once on the host."""
    f_max = float(sample_rate // 2)
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_pts = torch.linspace(0.0, 2595.0 * math.log10(1.0 + f_max / 700.0), n_mels + 2)
    f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    return torch.clamp(torch.min(-slopes[:, :-2] / f_diff[:-1], slopes[:, 2:] / f_diff[1:]), min=0.0).contiguous()


class _MelSpec(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, x: Tensor):
        x2 = _f32c(x.detach().reshape(-1, x.shape[-1]))
        cv, mel = mod._forward_raw(x2)
        ctx.mod, ctx.shape = mod, x.shape
        ctx.save_for_backward(cv)
        return mel.reshape(*x.shape[:-1], mod.n_mels, mel.shape[-1])

    @staticmethod
    def backward(ctx, g: Tensor):
        (cv,) = ctx.saved_tensors
        mod = ctx.mod
        dx = mod._backward_raw(cv, _f32c(g.reshape(-1, mod.n_mels, g.shape[-1])), ctx.shape[-1])
        return None, dx.reshape(ctx.shape)


class MelSpectrogram(nn.Module):
    """torchaudio.transforms.MelSpectrogram as the reference builds it: Hann window, centred / reflect-padded,
    power 2, ``normalized=True`` (window energy), HTK mel scale without area normalisation.
    ``hop_length`` must divide ``n_fft`` (it is ``win_length // 4`` in the reference)."""

    def __init__(self, sample_rate: int = 16000, n_fft: int = 400, win_length: Optional[int] = None,
                 hop_length: Optional[int] = None, n_mels: int = 128, normalized: bool = False):
        super().__init__()
        self.sample_rate, self.n_fft = sample_rate, n_fft
        self.win_length = n_fft if win_length is None else win_length
        self.hop_length = self.win_length // 2 if hop_length is None else hop_length
        self.n_mels, self.normalized = n_mels, normalized
        if n_fft % self.hop_length:
            raise NotImplementedError("the framed-DFT kernel needs hop_length | n_fft")
        self.register_buffer("fb", melscale_fbanks(n_fft // 2 + 1, sample_rate, n_mels), persistent=False)
        # taps of the framed DFT that meet the window (spectral.hip: fdft_geom) -- only those are executed
        left = (n_fft - self.win_length) // 2
        self._taps = -(-(left + self.win_length) // self.hop_length) - left // self.hop_length
        self._img = {}

    def _image(self, backward: int, device) -> Tensor:
        key = (backward, device)
        if key not in self._img:
            lib = _lib.load()
            n = lib.agx_fdft_packed_floats(self.n_fft, self.win_length, self.hop_length, 1, backward)
            if n < 0:
                _lib.check(int(n), "agx_fdft_packed_floats")
            img = torch.empty(int(n), dtype=torch.float32, device=device)
            _lib.check(lib.agx_fdft_pack(self.n_fft, self.win_length, self.hop_length, 1, 1, 2 if self.normalized else 0,
                                         backward, _ptr(img), _stream()), "agx_fdft_pack")
            self._img[key] = img
        return self._img[key]

    def _ws(self, b: int, length: int, device) -> Tensor:
        n = _lib.load().agx_fdft_workspace_bytes(b, length, self.n_fft, self.hop_length)
        if n < 0:
            _lib.check(int(n), "agx_fdft_workspace_bytes")
        return torch.empty(int(n) // 4 + 1, dtype=torch.float32, device=device)

    def _forward_raw(self, x: Tensor):
        lib = _lib.load()
        _need_gpu(x)
        b, length = x.shape
        t = lib.agx_fdft_frames(length, self.n_fft, self.hop_length)
        if t < 0:
            _lib.check(int(t), "agx_fdft_frames")
        rows = int(lib.agx_fdft_rows(self.n_fft, 1))
        cv = torch.empty(b, 2 * rows, int(t), dtype=torch.float32, device=x.device)
        ops.count_macs("fdft_forward", b * self._taps * self.hop_length * 2 * rows * int(t))
        _lib.check(lib.agx_fdft_forward(_ptr(x), _ptr(self._image(0, x.device)), _ptr(cv), _ptr(self._ws(b, length, x.device)),
                                        b, length, self.n_fft, self.win_length, self.hop_length, 1, _stream()),
                   "agx_fdft_forward")
        mel = torch.empty(b, self.n_mels, int(t), dtype=torch.float32, device=x.device)
        _lib.check(lib.agx_melpower(_ptr(cv), _ptr(self.fb), _ptr(mel), b, self.n_fft // 2 + 1, int(t), self.n_mels,
                                    _stream()), "agx_melpower")
        return cv, mel

    def _backward_raw(self, cv: Tensor, dmel: Tensor, length: int) -> Tensor:
        lib = _lib.load()
        b, _, t = cv.shape
        dcv = torch.empty_like(cv)
        _lib.check(lib.agx_melpower_backward(_ptr(cv), _ptr(self.fb), _ptr(dmel), _ptr(dcv), b, self.n_fft // 2 + 1, t,
                                             self.n_mels, _stream()), "agx_melpower_backward")
        dx = torch.empty(b, length, dtype=torch.float32, device=cv.device)
        ops.count_macs("fdft_backward", b * self._taps * self.hop_length * cv.shape[1] * t)
        _lib.check(lib.agx_fdft_backward(_ptr(dcv), _ptr(self._image(1, cv.device)), _ptr(dx),
                                         _ptr(self._ws(b, length, cv.device)), b, length, self.n_fft, self.win_length,
                                         self.hop_length, 1, _stream()), "agx_fdft_backward")
        return dx

    def forward(self, waveform: Tensor) -> Tensor:
        """(..., L) -> (..., n_mels, T)."""
        if torch.is_grad_enabled() and waveform.requires_grad:
            return _MelSpec.apply(self, waveform)
        x2 = _f32c(waveform.detach().reshape(-1, waveform.shape[-1]))
        _, mel = self._forward_raw(x2)
        return mel.reshape(*waveform.shape[:-1], self.n_mels, mel.shape[-1])


REDUCE_LOG_L2 = 5


def multispectral_reconstruction_loss(original: Tensor, reconstruction: Tensor, spectrograms: Sequence[MelSpectrogram],
                                      windows: Sequence[int] = tuple(2 ** i for i in range(5, 12)), eps: float = 1e-8,
                                      spec_loss_weight: float = 1, use_log_l2: bool = True, scale_alpha: bool = True):
    """training.py:51-78.  (``nan_to_num`` of the reference is a no-op on finite spectrograms and is not applied;
    ``eps`` is fixed at the reference's 1e-8 inside the log-L2 reduction kernel.)"""
    from .discriminator import _mean
    if eps != 1e-8:
        raise NotImplementedError("the log-L2 reduction kernel has eps = 1e-8 built in (the reference's value)")
    alphas = [math.sqrt(w / 2) if scale_alpha else 1.0 for w in windows]
    loss = 0
    for i, spec in enumerate(spectrograms):
        so = spec(original.detach())
        sr = spec(reconstruction)
        loss = loss + _mean(ops.REDUCE_L1, so, sr)
        if use_log_l2:
            loss = loss + alphas[i] * _mean(REDUCE_LOG_L2, so, sr)
        else:
            d = so - sr
            loss = loss + alphas[i] * (d * d).mean()
    return spec_loss_weight * loss
