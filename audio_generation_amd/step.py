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

``training_backward`` is the same micro-batch INCLUDING those two backward calls, scheduled for the GPU: the
``.grad`` it leaves on generator and discriminator parameters is the one the reference's two calls leave (their
sum on the discriminators -- the reference never clears the gradients ``loss.backward()`` adds there), but
each discriminator's graph is walked once and freed before the next one is built.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence, Tuple

import torch

from . import signal_ops as sg
from .discriminator import discriminator_generator_loss

Tensor = torch.Tensor


def _generator_side(model, x: Tensor, *, sample_rate, frequency_filter, codebook_frequency_scale, noise_aug_scale,
                    pre_emphasis, spectrograms, spec_windows, spec_loss_weight, reconstruction_loss_weight,
                    sparsity_weight, use_reconstruction_loss, use_commit_loss, update_codebook, prioritize_early,
                    codebook_n):
    """training.py:313-361: everything before the discriminators -> (x, y, loss, parts)."""
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
    return x, y, loss, parts


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
    x, y, loss, parts = _generator_side(
        model, x, sample_rate=sample_rate, frequency_filter=frequency_filter,
        codebook_frequency_scale=codebook_frequency_scale, noise_aug_scale=noise_aug_scale, pre_emphasis=pre_emphasis,
        spectrograms=spectrograms, spec_windows=spec_windows, spec_loss_weight=spec_loss_weight,
        reconstruction_loss_weight=reconstruction_loss_weight, sparsity_weight=sparsity_weight,
        use_reconstruction_loss=use_reconstruction_loss, use_commit_loss=use_commit_loss,
        update_codebook=update_codebook, prioritize_early=prioritize_early, codebook_n=codebook_n)
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


def training_backward(model, x: Tensor, discriminators: Sequence = (), *, generator_loss_weight: float = 1.0,
                      **terms) -> Tuple[Tensor, Optional[Tensor], Dict[str, float]]:
    """``training_losses`` + ``discriminator_loss.backward(retain_graph=True); loss.backward()``
    (``training.py:363-380``) with the same resulting ``.grad`` everywhere, in the order that suits the device:

    * the reconstruction ``y`` is cut into a leaf; every discriminator's two losses are differentiated in ONE
      autograd traversal (``torch.autograd.backward([d_i, g_i])``).  The reference's two calls walk the graph of
      the real-input pass twice -- once for the hinge term, once for the feature-matching term -- and ADD both
      results into the discriminator's ``.grad``; one traversal adds the upstream gradients first and runs the
      backward kernels of that pass once (4 -> 3 backward passes per discriminator).  The generator still only
      receives ``d loss / d y``: neither discriminator-loss term depends on it (``y.detach()``, real input).
    * each discriminator's graph (three passes' feature maps) is freed before the next one is built; the
      gradient it leaves on the leaf is accumulated and sent through the generator once at the end.

    Returns detached values ``(loss, discriminator_loss or None, parts)``; gradients are in ``.grad``."""
    x, y, loss, parts = _generator_side(
        model, x, **{**dict(sample_rate=24000, frequency_filter=None, codebook_frequency_scale=0.0, noise_aug_scale=0.0,
                            pre_emphasis=None, spectrograms=None, spec_windows=(), spec_loss_weight=1.0,
                            reconstruction_loss_weight=1.0, sparsity_weight=0.0, use_reconstruction_loss=True,
                            use_commit_loss=True, update_codebook=False, prioritize_early=False, codebook_n=None),
                     **terms})
    if not discriminators:
        if loss.requires_grad:
            loss.backward()
        return loss.detach(), None, {k: float(v.detach()) for k, v in parts.items()}
    y_leaf = y.detach().requires_grad_(True)
    total, d_total = loss.detach().clone(), y.new_zeros(())
    for disc in discriminators:
        g_i, d_i = discriminator_generator_loss(x, y_leaf, disc)
        parts[f"{disc.name}_g_loss"] = g_i.detach()
        torch.autograd.backward([d_i * generator_loss_weight, g_i * generator_loss_weight])
        total += g_i.detach() * generator_loss_weight
        d_total += d_i.detach()
        del g_i, d_i
    d_total = d_total * generator_loss_weight
    parts["discriminator_loss"] = d_total
    if loss.requires_grad:      # (every reconstruction-side term switched off: only the discriminators' gradient on y is left)
        torch.autograd.backward([loss, y], [torch.ones_like(loss), y_leaf.grad])
    else:
        y.backward(y_leaf.grad)
    return total, d_total, {k: float(v.detach()) for k, v in parts.items()}
