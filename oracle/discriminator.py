"""Oracle: the discriminators and their loss (CPU, functional torch).

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.  Restates ``networks/discriminator.py`` of the
reference on plain tensors + a state dict with the reference's keys
(``...weight_orig / weight_u / weight_v / bias``: old-style ``torch.nn.utils.spectral_norm``,
``utils.py:34-42``).  Pinned by ``tests/golden/g7_discriminators.npz``.

* ``spectral_weight``            -- ``spectral_norm`` forward pre-hook: one power iteration in training
                                    mode (buffers updated), none in eval; ``W / (u . W v)``.
* ``waveform_block``             -- ``WaveformDiscriminatorBlock.forward`` (discriminator.py:7-57)
* ``waveform_discriminator``     -- ``WaveFormDiscriminator.forward`` (:59-84)
* ``stft_two_sided``             -- the ``torch.stft`` call of ``STFTDiscriminator.forward`` (:181-187)
* ``stft_discriminator``         -- ``STFTDiscriminator.forward`` (:178-202)
* ``discriminator_generator_loss`` -- (:204-246)
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

WAVE_KERNELS = (15, 41, 41, 41, 41, 5, 3)
WAVE_STRIDES = (1, 4, 4, 4, 4, 1, 1)
WAVE_GROUPS = (1, 4, 16, 64, 256, 1, 1)
STFT_MULTIPLIERS = (2, 2, 1, 2, 1, 2)
STFT_STRIDES = ((1, 2), (2, 2)) * 3


def spectral_weight(sd: Dict[str, Tensor], prefix: str, train: bool, eps: float = 1e-12) -> Tensor:
    """``W_orig / sigma`` as torch's SpectralNorm.compute_weight does (dim 0, one power iteration
    when training); ``sd[prefix + 'weight_u' / 'weight_v']`` are replaced when training."""
    w = sd[prefix + "weight_orig"]
    u, v = sd[prefix + "weight_u"], sd[prefix + "weight_v"]
    mat = w.reshape(w.shape[0], -1)
    if train:
        with torch.no_grad():
            v = F.normalize(torch.mv(mat.t(), u), dim=0, eps=eps)
            u = F.normalize(torch.mv(mat, v), dim=0, eps=eps)
        sd[prefix + "weight_u"], sd[prefix + "weight_v"] = u, v
    sigma = torch.dot(u, torch.mv(mat, v))
    return w / sigma


def waveform_block(x: Tensor, sd: Dict[str, Tensor], prefix: str, scale: int, train: bool = False,
                   strides: Sequence[int] = WAVE_STRIDES, groups: Sequence[int] = WAVE_GROUPS,
                   apply_sigmoid: bool = True) -> Tuple[Tensor, List[Tensor]]:
    """AvgPool1d(2 scale, stride scale, padding scale) then 7 unpadded grouped convs, LeakyReLU(0.2)
    after all but the last; every layer output (the pooled input included) is a feature."""
    x = F.avg_pool1d(x, 2 * scale, stride=scale, padding=scale)
    feats = [x]
    n = len(strides)
    for i in range(n):
        last = i == n - 1
        p = f"{prefix}layers.{i + 1}." + ("" if last else "0.")
        w = spectral_weight(sd, p, train)
        x = F.conv1d(x, w, sd[p + "bias"], stride=strides[i], groups=groups[i])
        if not last:
            x = F.leaky_relu(x, 0.2)
        feats.append(x)
    return (torch.sigmoid(x) if apply_sigmoid else x), feats


def waveform_discriminator(x: Tensor, sd: Dict[str, Tensor], n_blocks: int = 3, factor: int = 2,
                           train: bool = False, prefix: str = "", **kw) -> Tuple[List[Tensor], List[Tensor]]:
    outs, feats = [], []
    for b in range(n_blocks):
        o, f = waveform_block(x, sd, f"{prefix}layers.{b}.", factor ** b, train, **kw)
        outs.append(o)
        feats.extend(f)
    return outs, feats


def stft_two_sided(x: Tensor, n_fft: int, hop: int, normalized: bool = True) -> Tensor:
    """``torch.stft(x, n_fft, hop, win_length=n_fft, normalized=..., onesided=False)`` with its defaults
    (rectangular window, center=True, reflect padding) -> real tensor (B, 2, T, F): the layout the
    reference rearranges to ("b f t c -> b c t f")."""
    pad = n_fft // 2
    xp = F.pad(x.unsqueeze(1), (pad, pad), mode="reflect").squeeze(1)
    frames = xp.unfold(-1, n_fft, hop)                       # (B, T, n_fft)
    spec = torch.fft.fft(frames, dim=-1)                      # (B, T, F)
    if normalized:
        spec = spec * (n_fft ** -0.5)
    return torch.stack([spec.real, spec.imag], dim=1)        # (B, 2, T, F)


def stft_discriminator(x: Tensor, sd: Dict[str, Tensor], win_length: int, train: bool = False,
                       multipliers: Sequence[int] = STFT_MULTIPLIERS,
                       strides: Sequence[Tuple[int, int]] = STFT_STRIDES, apply_sigmoid: bool = True,
                       normalize_stft: bool = True, prefix: str = "") -> Tuple[List[Tensor], List[Tensor]]:
    x = stft_two_sided(x.squeeze(1), win_length, win_length // 4, normalize_stft)
    x = F.conv2d(x, spectral_weight(sd, prefix + "first_conv.", train), sd[prefix + "first_conv.bias"], padding=3)
    feats = [x]
    for i, stride in enumerate(strides):
        p = f"{prefix}blocks.{i}.layers."
        x = F.leaky_relu(F.conv2d(x, spectral_weight(sd, p + "0.", train), sd[p + "0.bias"], padding=1), 0.2)
        k = (stride[0] + 2, stride[1] + 2)
        x = F.conv2d(x, spectral_weight(sd, p + "2.", train), sd[p + "2.bias"], stride=stride,
                     padding=((k[0] - 1) // 2, (k[1] - 1) // 2))
        feats.append(x)
    fk = win_length // (2 ** (len(multipliers) + 1))
    x = F.conv2d(x, spectral_weight(sd, prefix + "final_conv.", train), sd[prefix + "final_conv.bias"],
                 padding=(0, (fk - 1) // 2))
    return [torch.sigmoid(x) if apply_sigmoid else x], feats


def discriminator_generator_loss(original: Tensor, reconstruction: Tensor,
                                 disc: Callable[[Tensor], Tuple[List[Tensor], List[Tensor]]],
                                 feature_multiplier: float = 100, scale_feature_loss: bool = True
                                 ) -> Tuple[Tensor, Tensor]:
    """discriminator.py:204-246: three passes through D (real, fake, fake detached), hinge losses
    averaged over the D outputs, L1 feature matching (optionally scaled by mean|x + 1e-3|)."""
    original_d, original_f = disc(original)
    recon_d, recon_f = disc(reconstruction)
    recon_d2, _ = disc(reconstruction.detach())
    k = len(original_d)
    d_loss, g_loss = 0, 0
    for x, y, y2 in zip(original_d, recon_d, recon_d2):
        real = -torch.minimum(x - 1, torch.zeros_like(x)).mean()
        fake = -torch.minimum(-y2 - 1, torch.zeros_like(y2)).mean()
        d_loss = d_loss + (real + fake) / k
        g_loss = g_loss - y.mean() / k
    f_loss, n = 0, len(original_f)
    for x, y in zip(original_f, recon_f):
        li = F.l1_loss(x, y) / n
        if scale_feature_loss:
            li = li / torch.abs(x + 1e-3).mean()
        f_loss = f_loss + li
    return g_loss + feature_multiplier * f_loss, d_loss
