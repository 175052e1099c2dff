"""Drop-in wavelet / multiresolution layers executed by libagx.

Mirrors ``networks/wavelets.py``: ``CausalMultiresConv1d`` (:38-96),
``MultiresScaleBlock`` (:98-121) and ``WaveletLayer`` (:123-234) -- same
constructor arguments, parameter / buffer names (``h0``, ``h1``, ``w``;
``conv_in.*``, ``conv_out.*``, ``space``, ``wavelet_scale``, ``cos_kernel``).

* the multires cascade is ONE kernel (``agx_multires_forward``): the 2*depth
  depthwise convs, the per-channel mixing and the GELU never touch HBM in between;
* the wavelet layer is conv_in (same-padded conv kernel) -> ``agx_wavelet_fold``
  (the (B,C,L,n_points) expansion of the reference is never materialised: the
  sliding-window sum collapses to two taps with per-channel phase sums) ->
  conv_out with the decoder block's activation fused.
"""
from __future__ import annotations

from math import sqrt
from typing import Optional

import torch
from torch import nn

from . import ops
from ._lib import CONV_CAUSAL, CONV_SAME, EPI_GELU_PRE, EPI_LEAKY_PRE, needs_grad

Tensor = torch.Tensor


class CausalMultiresConv1d(nn.Module):
    """wavelets.py:38-96."""

    def __init__(self, channels, kernel_size, depth, dropout=0.0, activation=None):
        super().__init__()
        if activation is not None and not isinstance(activation, nn.GELU):
            raise NotImplementedError("only GELU is fused into the multires kernel")
        if dropout != 0.0:
            raise NotImplementedError("dropout > 0 is training-only and not on the forward path")
        self.channels, self.kernel_size, self.depth, self.dropout = channels, kernel_size, depth, dropout
        self.activation = nn.GELU() if activation is None else activation
        scalar = sqrt(2.0) / (kernel_size * 2)
        self.h0 = nn.Parameter(torch.empty(channels, 1, kernel_size).uniform_(-1., 1.) * scalar)
        self.h1 = nn.Parameter(torch.empty(channels, 1, kernel_size).uniform_(-1., 1.) * scalar)
        self.w = nn.Parameter(torch.empty(channels, depth + 2).uniform_(-1., 1.) * sqrt(2.0 / (2 * depth + 4)))
        self.dropout_layer = nn.Dropout(dropout)

    def _hip(self, x: Tensor) -> Tensor:
        return ops.multires_forward(x, self.h0.detach(), self.h1.detach(), self.w.detach(), self.depth)

    def forward(self, x: Tensor) -> Tensor:
        if needs_grad(x, self):
            return _MultiresNative.apply(self, x, self.h0, self.h1, self.w)
        return self._hip(x)


class _MultiresNative(torch.autograd.Function):
    """``agx_multires_forward`` / ``agx_multires_backward`` (the cascade is re-formed per tile in the backward kernel:
    only the layer input is kept)."""

    @staticmethod
    def forward(ctx, mod: "CausalMultiresConv1d", x: Tensor, h0: Tensor, h1: Tensor, w: Tensor):
        ctx.depth = mod.depth
        ctx.save_for_backward(x.detach(), h0.detach(), h1.detach(), w.detach())
        with torch.no_grad():
            return mod._hip(x.detach())

    @staticmethod
    def backward(ctx, g: Tensor):
        x, h0, h1, w = ctx.saved_tensors
        dx, dh0, dh1, dw = ops.multires_backward(x, g.contiguous(), h0, h1, w, ctx.depth)
        return None, dx, dh0, dh1, dw


class _PlainConv(nn.Module):
    """Parameter holder standing where the reference has a plain ``torch.nn.Conv1d``
    (keys ``weight`` / ``bias``), with torch's default init."""

    def __init__(self, c_in, c_out, kernel):
        super().__init__()
        ref = nn.Conv1d(c_in, c_out, kernel)
        self.weight, self.bias = ref.weight, ref.bias
        self.in_channels, self.out_channels, self.kernel_size = c_in, c_out, (kernel,)
        self._key, self._packed = None, None
        self._bkey, self._packed_bwd = None, None

    def desc(self, kind: int, x: Tensor, epilogue: int = 0, slope: float = 0.1):
        return ops.conv_desc(kind, x.shape[0], self.in_channels, self.out_channels, x.shape[2],
                             self.kernel_size[0], 1, 1, epilogue, slope)

    def packed_bwd(self, kind: int) -> Tensor:
        key = (kind, self.weight.data_ptr(), self.weight._version)
        if key != self._bkey:
            d = ops.conv_desc(kind, 1, self.in_channels, self.out_channels, 1 << 20, self.kernel_size[0])
            self._packed_bwd, self._bkey = ops.conv_pack_bwd(d, self.weight.detach()), key
        return self._packed_bwd

    def run(self, x: Tensor, kind: int, epilogue: int = 0, slope: float = 0.1) -> Tensor:
        key = (kind, self.weight.data_ptr(), self.weight._version)
        if key != self._key:
            d = ops.conv_desc(kind, 1, self.in_channels, self.out_channels, 1 << 20, self.kernel_size[0])
            self._packed, self._key = ops.conv_pack(d, self.weight.detach()), key
        desc = ops.conv_desc(kind, x.shape[0], self.in_channels, self.out_channels, x.shape[2],
                             self.kernel_size[0], 1, 1, epilogue, slope)
        return ops.conv_forward(desc, x, self._packed, self.bias.detach())


class MultiresScaleBlock(nn.Module):
    """wavelets.py:98-121: multires -> nearest upsample -> 1x1 conv -> GELU.  The
    1x1 conv commutes with the nearest upsample, so it runs on the low-rate
    signal with the GELU fused and the repeat is a plain copy afterwards."""

    def __init__(self, in_channels, out_channels, scale_factor=2, kernel_size=3, multires_depth=6,
                 dropout=0.0, activation=None):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.scale_factor, self.kernel_size = scale_factor, kernel_size
        self.activation = nn.GELU() if activation is None else activation
        self.multires_conv = CausalMultiresConv1d(in_channels, kernel_size, multires_depth, dropout, activation)
        self.conv = _PlainConv(in_channels, out_channels, 1)

    def _hip(self, x: Tensor) -> Tensor:
        y = self.conv.run(self.multires_conv._hip(x), CONV_CAUSAL, EPI_GELU_PRE)
        return y.repeat_interleave(self.scale_factor, dim=-1)          # a copy: the 1x1 conv ran on the low rate

    def forward(self, x: Tensor) -> Tensor:
        if needs_grad(x, self):
            mr = self.multires_conv
            return _ScaleBlockNative.apply(self, x, mr.h0, mr.h1, mr.w, self.conv.weight, self.conv.bias)
        return self._hip(x)


class _ScaleBlockNative(torch.autograd.Function):
    """Backward of ``MultiresScaleBlock`` on the HIP kernels: the nearest upsample's adjoint (sum of each group of
    ``scale_factor`` gradients) fused with the exact-GELU gradient (``agx_group_sum``), the k = 1 conv's backward-data /
    weight-gradient kernels, ``agx_multires_backward``.  The two hidden tensors are re-materialised."""

    @staticmethod
    def forward(ctx, blk: "MultiresScaleBlock", x: Tensor, *params: Tensor):
        ctx.blk = blk
        ctx.save_for_backward(x.detach())
        with torch.no_grad():
            return blk._hip(x.detach())

    @staticmethod
    def backward(ctx, g: Tensor):
        blk, (x,) = ctx.blk, ctx.saved_tensors
        mr, conv = blk.multires_conv, blk.conv
        h0, h1, w = mr.h0.detach(), mr.h1.detach(), mr.w.detach()
        m = ops.multires_forward(x, h0, h1, w, mr.depth)
        pre = conv.run(m, CONV_CAUSAL)                                  # pre-activation of the 1x1 conv
        dpre = ops.group_sum(g.contiguous(), blk.scale_factor, gelu_pre=pre)
        desc = conv.desc(CONV_CAUSAL, m)
        dwc, _, dbc = ops.conv_bwd_weight(desc, m, dpre, conv.weight.detach(), None)
        dm = ops.conv_bwd_data(desc, dpre, conv.packed_bwd(CONV_CAUSAL))
        dx, dh0, dh1, dw = ops.multires_backward(x, dm, h0, h1, w, mr.depth)
        return None, dx, dh0, dh1, dw, dwc, dbc


class WaveletLayer(nn.Module):
    """wavelets.py:123-234 (``multires_depth = 0``, the only wiring the model uses,
    vae.py:167-173)."""

    def __init__(self, in_channels, hidden_channels, out_channels=None, wavelet_kernel_size=13,
                 out_conv_kernel_size=3, scale_factor=2, n_points=16, interval=(-10, 10), wavelet_scale=40,
                 multires_depth=0, channelwise_scale=True):
        super().__init__()
        assert n_points % scale_factor == 0, "n_points must be divisible by scale_factor"
        if multires_depth > 0:
            raise NotImplementedError("WaveletLayer(multires_depth > 0) feeds a 4-D tensor to a 1-D conv in the "
                                      "reference (wavelets.py:215-219) and cannot run there either")
        self.in_channels = in_channels
        self.out_channels = in_channels if out_channels is None else out_channels
        self.hidden_channels = hidden_channels
        self.wavelet_kernel_size, self.out_conv_kernel_size = wavelet_kernel_size, out_conv_kernel_size
        self.n_points, self.scale_factor = n_points, scale_factor
        self.fold_dim = n_points // scale_factor
        self.multires = False
        self.conv_in = _PlainConv(in_channels, hidden_channels, wavelet_kernel_size)
        self.conv_out = _PlainConv(hidden_channels, self.out_channels, out_conv_kernel_size)
        self.register_buffer("space", torch.linspace(*interval, n_points).reshape(1, 1, 1, n_points))
        scale = torch.tensor(wavelet_scale).float()
        if channelwise_scale:
            scale = scale.repeat(hidden_channels).reshape(1, hidden_channels, 1, 1)
        self.wavelet_scale = nn.Parameter(scale)
        self.register_buffer("cos_kernel", torch.cos(self.space))

    def run_fused(self, x: Tensor, post_slope: Optional[float] = None) -> Tensor:
        h = self.conv_in.run(x, CONV_SAME)
        y = ops.wavelet_fold(h, self.space, self.wavelet_scale.detach(), self.scale_factor)
        return self.conv_out.run(y, CONV_SAME, EPI_LEAKY_PRE if post_slope is not None else 0, post_slope or 0.0)

    def forward(self, x: Tensor) -> Tensor:
        return self.run_fused(x, None)

    # -- native backward (used by native_backward.py as one unit of a decoder stack) --------------
    def params(self):
        return [self.conv_in.weight, self.conv_in.bias, self.conv_out.weight, self.conv_out.bias, self.wavelet_scale]

    def backward_native(self, x: Tensor, dz: Tensor, mask: Optional[Tensor], mask_slope: float):
        """``dz`` = gradient w.r.t. conv_out's linear output.  Returns (dx, grads in ``params()`` order);
        the hidden tensors are re-materialised (two cheap kernels) rather than kept from the forward."""
        h = self.conv_in.run(x, CONV_SAME)
        y = ops.wavelet_fold(h, self.space, self.wavelet_scale.detach(), self.scale_factor)
        dwo, _, dbo = ops.conv_bwd_weight(self.conv_out.desc(CONV_SAME, y), y, dz, self.conv_out.weight.detach(), None)
        dy = ops.conv_bwd_data(self.conv_out.desc(CONV_SAME, y), dz, self.conv_out.packed_bwd(CONV_SAME))
        dh, dsig = ops.wavelet_fold_backward(h, dy, self.space, self.wavelet_scale.detach(), self.scale_factor)
        dwi, _, dbi = ops.conv_bwd_weight(self.conv_in.desc(CONV_SAME, x), x, dh, self.conv_in.weight.detach(), None)
        dx = ops.conv_bwd_data(self.conv_in.desc(CONV_SAME, x), dh, self.conv_in.packed_bwd(CONV_SAME), None, mask,
                               mask_slope)
        return dx, [dwi, dbi, dwo, dbo, dsig]
