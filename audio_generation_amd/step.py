"""One micro-batch of the reference's training step on the HIP path.

Mirror of the body of ``Trainer.mini_epoch``'s inner loop (``networks/training.py:313-376``) -- the
caller of the codec forward that BASELINE config 5 names: low-pass of the input batch (``:313-318``),
optional noise augmentation (``:320-323``), ``model(x_, update_codebook, prioritize_early,
codebook_n)`` (``:325-328``), pre-emphasised MSE (``:330-341``), commitment loss (``:345-347``),
sparsity term (``:350-353``), 7-window mel loss (``:355-361``) and one
``discriminator_generator_loss`` per discriminator (``:363-374``).  As in the reference, the
pre-emphasised pair ``(x, y)`` replaces the raw one for every later term.  The loop around it (data
loader, accumulation, optimizers, bookkeeping, ``:296-311, 376-390``) stays the caller's: this function
only composes kernels-backed ops and returns the two losses the caller calls ``backward()`` on --
``discriminator_loss.backward(retain_graph=True)`` first, then ``loss.backward()`` (``:374, 380``).
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence, Tuple

import torch

from . import signal_ops as sg
from .discriminator import discriminator_generator_loss

Tensor = torch.Tensor


def training_losses(model, x: Tensor, discriminators: Sequence = (), *, sample_rate: int = 24000,
                    frequency_filter: Optional[float] = None, codebook_frequency_scale: float = 0.0,
                    noise_aug_scale: float = 0.0, pre_emphasis: Optional[float] = None,
                    spectrograms: Optional[Sequence] = None, spec_windows: Sequence[int] = (),
                    spec_loss_weight: float = 1.0, reconstruction_loss_weight: float = 1.0,
                    generator_loss_weight: float = 1.0, sparsity_weight: float = 0.0,
                    use_reconstruction_loss: bool = True, use_commit_loss: bool = True,
                    update_codebook: bool = False, prioritize_early: bool = False, codebook_n=None
                    ) -> Tuple[Tensor, Optional[Tensor], Dict[str, float]]:
    """-> (generator-side loss, summed discriminator loss or None, {term: value})."""
    parts: Dict[str, Tensor] = {}
    if frequency_filter is not None:                                               # training.py:313-318
        n = model.num_quantizers if codebook_n is None else codebook_n
        x = sg.lowpass_biquad(x, sample_rate, frequency_filter * (1 + n * codebook_frequency_scale))
    x_in = x + torch.randn_like(x) * noise_aug_scale if noise_aug_scale else x    # :320-323
    y, commit_loss, _ = model(x_in, update_codebook=update_codebook, prioritize_early=prioritize_early,
                              codebook_n=codebook_n)                               # :325-328
    loss = y.new_zeros(())
    if use_reconstruction_loss:                                                    # :330-341
        if pre_emphasis is not None:
            x, y = sg.preemphasis(x, pre_emphasis), sg.preemphasis(y, pre_emphasis)
        parts["reconstruction_loss"] = ((x - y) ** 2).mean() * reconstruction_loss_weight
        loss = loss + parts["reconstruction_loss"]
    if use_commit_loss:                                                            # :345-347
        parts["commit_loss"] = commit_loss
        loss = loss + commit_loss
    if sparsity_weight > 0:                                                        # :350-353
        parts["sparsity_loss"] = sparsity_weight * y.abs().mean()
        loss = loss + parts["sparsity_loss"]
    if spectrograms:                                                               # :355-361
        parts["multispectral_loss"] = sg.multispectral_reconstruction_loss(
            x, y, spectrograms, spec_windows, spec_loss_weight=spec_loss_weight)
        loss = loss + parts["multispectral_loss"]
    d_loss = None
    if discriminators:                                                             # :363-373
        d_loss = y.new_zeros(())
        for disc in discriminators:
            g_i, d_i = discriminator_generator_loss(x, y, disc)
            parts[f"{disc.name}_g_loss"] = g_i
            loss = loss + g_i * generator_loss_weight
            d_loss = d_loss + d_i
        d_loss = d_loss * generator_loss_weight
        parts["discriminator_loss"] = d_loss
    return loss, d_loss, {k: float(v.detach()) for k, v in parts.items()}
