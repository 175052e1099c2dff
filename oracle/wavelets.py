"""Oracle: multiresolution cascade and wavelet upsampling layer (CPU, torch fp32).

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.
Follows ``networks/wavelets.py``: ``causal_functional_conv1d`` (:8-26),
``CausalMultiresConv1d.forward`` (:79-96), ``MultiresScaleBlock.forward``
(:117-121), ``WaveletLayer.forward`` (:213-234).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

from .codec import causal_pads

Tensor = torch.Tensor


def causal_depthwise(x: Tensor, h: Tensor, dilation: int) -> Tensor:
    """Depthwise causal conv: ``h`` is (C,1,K); left pad ``dilation*(K-1)``
    (wavelets.py:17-26 with stride 1, groups = C)."""
    left, right = causal_pads(x.shape[-1], h.shape[-1], 1, dilation)
    return F.conv1d(F.pad(x, (left, right)), h, None, dilation=dilation, groups=x.shape[1])


def multires_conv(x: Tensor, h0: Tensor, h1: Tensor, w: Tensor, depth: int) -> Tensor:
    """wavelets.py:79-96.  ``w`` is (C, depth+2): column 0 weights the final
    low-pass residual, columns 1..depth the high-pass outputs (deepest level
    first in time: the loop runs i = depth..1 while the dilation doubles),
    column -1 the input itself.  Exact (erf) GELU at the end; dropout p=0."""
    low = x
    y = torch.zeros_like(x)
    dilation = 1
    for i in range(depth, 0, -1):
        high = causal_depthwise(low, h1, dilation)
        low = causal_depthwise(low, h0, dilation)
        y = y + w[:, i:i + 1] * high
        dilation *= 2
    y = y + w[:, :1] * low
    y = y + x * w[:, -1:]
    return F.gelu(y)


def multires_scale_block(x: Tensor, h0: Tensor, h1: Tensor, w: Tensor, depth: int,
                         conv_w: Tensor, conv_b: Optional[Tensor], scale_factor: int) -> Tensor:
    """wavelets.py:117-121: multires -> nearest upsample -> 1x1 conv -> GELU."""
    y = multires_conv(x, h0, h1, w, depth)
    y = y.repeat_interleave(scale_factor, dim=-1)
    return F.gelu(F.conv1d(y, conv_w, conv_b))


def wavelet_fold(h: Tensor, space: Tensor, wavelet_scale: Tensor, scale_factor: int) -> Tensor:
    """The middle of ``WaveletLayer.forward`` (wavelets.py:221-231).

    ``h`` (B,C,L) -> (B,C,L*scale_factor).  Every input step emits a
    ``n_points`` long wavelet ``cos(t) * exp(-t^2 / sigma_c) * h`` laid end to
    end; the output is the sliding-window sum (window ``n_points``, hop
    ``fold = n_points // scale_factor``) of that flat signal.  The window count
    comes up ``scale_factor - 1`` short of ``L*scale_factor``; the reference
    then appends the last ``scale_factor - 1`` raw samples of the flat signal
    (NOT window sums) -- reproduced verbatim.
    """
    n_points = space.numel()
    fold = n_points // scale_factor
    t = space.reshape(1, 1, 1, n_points)
    sigma = wavelet_scale if wavelet_scale.dim() == 4 else wavelet_scale.reshape(1, 1, 1, 1)
    kernel = torch.cos(t) * torch.exp(-(t ** 2) / sigma)
    flat = (kernel * h.unsqueeze(-1)).flatten(2)             # (B,C,L*n_points)
    expected = flat.shape[-1] // fold
    out = flat.unfold(-1, n_points, fold).sum(dim=-1)
    short = out.shape[-1] - expected
    if short < 0:
        out = torch.cat([out, flat[..., short:]], dim=-1)
    return out


def wavelet_layer(x: Tensor, sd: Dict[str, Tensor], prefix: str, scale_factor: int) -> Tensor:
    """``WaveletLayer.forward`` with ``multires_depth=0`` (wavelets.py:213-234):
    same-padded conv -> wavelet fold -> same-padded conv.  Parameters are read
    from ``sd`` under ``prefix`` (``conv_in.*``, ``conv_out.*``, ``space``,
    ``wavelet_scale``)."""
    h = F.conv1d(x, sd[prefix + "conv_in.weight"], sd.get(prefix + "conv_in.bias"), padding="same")
    y = wavelet_fold(h, sd[prefix + "space"], sd[prefix + "wavelet_scale"], scale_factor)
    return F.conv1d(y, sd[prefix + "conv_out.weight"], sd.get(prefix + "conv_out.bias"), padding="same")
