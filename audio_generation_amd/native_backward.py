"""Native backward of the conv stacks on the HIP kernels.

A stack (``model.encoders`` / ``model.decoders``) is flattened into *units*:

* conv unit      ``y = act(conv(x) + b)``             (act optional)
* residual unit  ``y = act(x + conv2(act(conv1(x) + b1)) + b2)``
* depthwise residual unit  ``y = act(x + conv2(act(conv1(dw(x)) + b1)) + b2)`` with ``dw`` the per-channel k = 1 conv
  of ``vae.py:103-105`` (grouped-conv backward kernels of ``conv_grouped_bwd.hip``)
* wavelet unit   ``y = act(conv_out(fold(conv_in(x))))``  (``WaveletLayer.backward_native``)

Forward runs the usual fused kernels and keeps every unit's input.  Backward walks the
units in reverse with three C-ABI ops per conv:

* ``agx_conv_bwd_data``   -- gradient w.r.t. the unit input; the residual branch is added and
  the LeakyReLU gradient of the *producing* unit is applied in the same kernel's epilogue
  (mask = the saved activation), so the gradient that travels between units is always the
  one w.r.t. the pre-activation;
* ``agx_conv_bwd_weight`` -- dv / dg / dbias (weight-norm chain rule included);
* (``SAVE_HIDDEN = False`` only) one ``agx_conv_forward`` per residual unit to re-materialise the hidden activation the
  fused forward kernel never wrote.

A stack containing a layer without backward kernels (an activation the kernels do not fuse) raises ``AgxError``:
there is no ATen fallback.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
from torch import nn

from . import ops
from ._lib import CONV_CAUSAL, EPI_LEAKY_PRE, AgxError

# Training forward of a residual unit: True = conv1 and conv2 as two launches that leave the hidden activation in
# HBM for the backward (one C x L tensor per block: +3.5 GB at config S, batch 32); False = the fused inference
# kernel, and the backward re-materialises the hidden activation with one more conv launch (6 % slower step).
SAVE_HIDDEN = True

Tensor = torch.Tensor


class _Unit:
    """One conv or residual unit of a stack."""

    def __init__(self, kind: str, convs: Sequence[nn.Module], slope: Optional[float], inner_slope: Optional[float] = None):
        self.kind = kind            # "conv" | "res" | "resdw" | "wavelet" | "multires"
        self.convs = list(convs)    # [_ConvBase], [conv1, conv2] or [depthwise, conv1, conv2]
        self.slope = slope          # activation after the unit (None = linear output)
        self.inner_slope = inner_slope

    def params(self) -> List[Tensor]:
        if self.kind == "wavelet":
            return self.convs[0].params()
        if self.kind == "multires":
            m = self.convs[0]
            return [m.h0, m.h1, m.w]
        out = []
        for c in self.convs:
            cp = c.conv
            out += [cp.weight_v, cp.weight_g] if hasattr(cp, "weight_v") else [cp.weight]
            if cp.bias is not None:
                out.append(cp.bias)
        return out


def build_units(stack: nn.ModuleList) -> List[_Unit]:
    """Flatten a ``CausalVQAE`` encoder / decoder ModuleList; raises if some layer has no backward kernels."""
    from .vae import (CausalDecoderBlock, CausalEncoderBlock, CausalResidualBlock1d, _ConvBase, _leaky_slope)

    units: List[_Unit] = []

    def add(layer, act):
        slope = _leaky_slope(act) if act is not None else None
        if isinstance(layer, CausalResidualBlock1d):
            inner = _leaky_slope(layer.activation)
            if inner is None:
                return False
            if getattr(layer, "depthwise", False):
                units.append(_Unit("resdw", [layer.conv1[0], layer.conv1[1], layer.conv2], slope, inner))
            else:
                units.append(_Unit("res", [layer.conv1, layer.conv2], slope, inner))
            return True
        if isinstance(layer, _ConvBase):
            units.append(_Unit("conv", [layer], slope))
            return True
        if hasattr(layer, "backward_native"):                 # WaveletLayer
            units.append(_Unit("wavelet", [layer], slope))
            return True
        return False

    for m in stack:
        if isinstance(m, nn.Sequential) and len(m) == 2 and isinstance(m[0], nn.Identity):   # encoders[0]
            ok = add(m[1], None)
        elif isinstance(m, CausalEncoderBlock):
            ok = all(add(seq[0], seq[1]) for seq in m.layers)
            if ok and hasattr(m, "multires"):      # build-defined placement: behind the strided conv (whose pair has no activation)
                units.append(_Unit("multires", [m.multires], None))
        elif isinstance(m, CausalDecoderBlock):
            ok = add(m.in_conv[0], m.in_conv[1])
            if ok and hasattr(m, "multires"):
                units.append(_Unit("multires", [m.multires], None))
            ok = ok and all(add(seq[0], seq[1]) for seq in m.layers)
        else:
            ok = add(m, None)
        if not ok:
            raise AgxError(f"no backward kernels for a layer of {type(m).__name__} (activation or layer type the HIP "
                           "kernels do not cover); there is no ATen fallback")
    return units


def _desc(conv, x: Tensor, epilogue: int = 0, slope: float = 0.1):
    c = conv.conv
    return ops.conv_desc(conv.kind, x.shape[0], c.in_channels, c.out_channels, x.shape[2], c.kernel_size[0],
                         c.stride[0], c.dilation[0], epilogue, slope, conv.impl)


def _vg(conv):
    cp = conv.conv
    return (cp.weight_v.detach(), cp.weight_g.detach()) if hasattr(cp, "weight_v") else (cp.weight.detach(), None)


def _grads_of(conv, x: Tensor, dz: Tensor) -> List[Tensor]:
    """[dv, dg, dbias] / [dweight, dbias] in the order of ``_Unit.params``."""
    v, g = _vg(conv)
    dv, dg, db = ops.conv_bwd_weight(_desc(conv, x), x, dz, v, g, want_bias=conv.conv.bias is not None)
    out = [dv] if g is None else [dv, dg]
    if db is not None:
        out.append(db)
    return out


def _dw_weight(conv) -> Tensor:
    """Effective (C, 1, 1) weight of the weight-normed per-channel conv: g v / |v| (a C-element parameter fold)."""
    v, g = _vg(conv)
    return v if g is None else v * (g / v.abs().clamp_min(1e-30))


def _dw_desc(conv, x: Tensor):
    c = conv.conv
    return ops.conv_desc(conv.kind, x.shape[0], c.in_channels, c.out_channels, x.shape[2], 1, 1, 1, groups=c.groups)


def _dw_grads(conv, x: Tensor, du: Tensor) -> List[Tensor]:
    """[dv, dg, dbias] / [dweight, dbias] of the per-channel conv: ``agx_conv_grouped_bwd_weight`` + the weight-norm chain
    rule on the C-element parameter vectors (for a 1-element direction v: dg = dw sign(v), dv = 0 up to rounding)."""
    v, g = _vg(conv)
    dw, db = ops.conv_grouped_bwd_weight(_dw_desc(conv, x), x, du, want_bias=conv.conv.bias is not None)
    if g is None:
        out = [dw]
    else:
        n = v.abs().clamp_min(1e-30)
        dot = dw * v
        out = [(g / n) * (dw - v * dot / (n * n)), dot / n]
    if db is not None:
        out.append(db)
    return out


class _NativeStack(torch.autograd.Function):
    @staticmethod
    def forward(ctx, units: List[_Unit], x: Tensor, *params: Tensor):
        from .vae import _run_unit_forward
        ctx.units = units
        inputs, hidden = [], []
        with torch.no_grad():
            for u in units:
                inputs.append(x)
                if u.kind == "res" and SAVE_HIDDEN:
                    # conv1 and conv2 as two launches: the hidden activation the backward needs is written once
                    # here instead of being re-materialised there (the fused inference kernel never writes it)
                    c1, c2 = u.convs
                    h = c1.run(x, EPI_LEAKY_PRE, u.inner_slope)
                    epi = ops.EPI_RESIDUAL | (ops.EPI_LEAKY_POST if u.slope is not None else 0)
                    x = c2.run(h, epi, u.slope or 0.0, res=x)
                    hidden.append(h)
                else:
                    x = _run_unit_forward(u, x)
        ctx.n_inputs = len(inputs)
        ctx.save_for_backward(*inputs, *hidden)
        ctx.mark_non_differentiable()
        return x

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        units, inputs = ctx.units, ctx.saved_tensors[:ctx.n_inputs]
        hidden = list(ctx.saved_tensors[ctx.n_inputs:])
        if units[-1].slope is not None:
            raise NotImplementedError("a stack ending in an activation needs its output saved for the mask")
        dz = grad_out.contiguous()                    # gradient w.r.t. the last unit's linear output
        grads: List[List[Tensor]] = [None] * len(units)
        for i in range(len(units) - 1, -1, -1):
            u, x = units[i], inputs[i]
            # the unit that produced x (if any) applied an activation: fold its gradient into dx
            prev_slope = units[i - 1].slope if i > 0 else None
            mask = x if prev_slope is not None else None
            if u.kind == "wavelet":
                dz, grads[i] = u.convs[0].backward_native(x, dz, mask, prev_slope or 0.0)
            elif u.kind == "multires":
                if mask is not None:
                    raise AgxError("a multires unit behind an activation has no fused mask (the blocks place it behind a linear conv)")
                m = u.convs[0]
                dz, dh0, dh1, dw = ops.multires_backward(x, dz, m.h0.detach(), m.h1.detach(), m.w.detach(), m.depth)
                grads[i] = [dh0, dh1, dw]
            elif u.kind == "conv":
                conv = u.convs[0]
                grads[i] = _grads_of(conv, x, dz)
                dz = ops.conv_bwd_data(_desc(conv, x), dz, conv.conv.packed_bwd(conv.kind), None, mask,
                                       prev_slope or 0.0)
            elif u.kind == "resdw":
                dwc, c1, c2 = u.convs
                ux = dwc.run(x)                                           # re-materialised: one bandwidth-bound launch
                h = c1.run(ux, EPI_LEAKY_PRE, u.inner_slope)
                g2 = _grads_of(c2, h, dz)
                dh = ops.conv_bwd_data(_desc(c2, h), dz, c2.conv.packed_bwd(CONV_CAUSAL), None, h, u.inner_slope)
                g1 = _grads_of(c1, ux, dh)
                du = ops.conv_bwd_data(_desc(c1, ux), dh, c1.conv.packed_bwd(CONV_CAUSAL))
                g0 = _dw_grads(dwc, x, du)
                dz = ops.conv_grouped_bwd_data(_dw_desc(dwc, x), du, _dw_weight(dwc), None, dz, mask, prev_slope or 0.0)
                grads[i] = g0 + g1 + g2
            else:
                c1, c2 = u.convs
                # the hidden activation: saved by the forward, or re-materialised with one conv launch
                h = hidden.pop() if hidden else c1.run(x, EPI_LEAKY_PRE, u.inner_slope)
                g2 = _grads_of(c2, h, dz)
                dh = ops.conv_bwd_data(_desc(c2, h), dz, c2.conv.packed_bwd(CONV_CAUSAL), None, h, u.inner_slope)
                g1 = _grads_of(c1, x, dh)
                dz = ops.conv_bwd_data(_desc(c1, x), dh, c1.conv.packed_bwd(CONV_CAUSAL), dz, mask,
                                       prev_slope or 0.0)
                grads[i] = g1 + g2
        flat = [g for gl in grads for g in gl]
        return (None, dz, *flat)


def run_stack(units: List[_Unit], x: Tensor) -> Tensor:
    params = [p for u in units for p in u.params()]
    return _NativeStack.apply(units, x, *params)
