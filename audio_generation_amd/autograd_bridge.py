"""Fallback trainability: HIP forward, ATen backward -- FENCED: it raises unless ``AGX_ALLOW_ATEN_BRIDGE=1``.

Every default path has hand-written backward kernels now (native_backward.py, transformers.py,
discriminator.py); this bridge is what remains for the shapes they do not cover (attention head_dim > 128,
discriminators with an activation other than LeakyReLU, encoder stacks with norm != Identity).  Such a module runs

* **forward** through libagx exactly as in inference (no autograd graph), and
* **backward** by re-evaluating an ATen restatement of the same module
  (``module._aten(x)``: ``F.conv1d`` & friends on the GPU) under
  ``torch.enable_grad`` and differentiating that.

This is clearly NOT the measured path: ``bench.py`` and every parity test run
under ``torch.no_grad()`` and never touch this file's ATen code.  It exists so
that the reference's training loop (``training.py:325-385``) can already drive
the drop-in modules; gradients are checked against the oracle's autograd in
``tests/test_gpu_training.py``.
"""
from __future__ import annotations

import os
from typing import Callable, Sequence

import torch

from ._lib import AgxError

Tensor = torch.Tensor


def require_allowed(what: str) -> None:
    """The bridge differentiates an ATen restatement (MIOpen / rocBLAS kernels, not this library's).  It must
    never be reached silently: opt in with ``AGX_ALLOW_ATEN_BRIDGE=1``."""
    if os.environ.get("AGX_ALLOW_ATEN_BRIDGE", "0") != "1":
        raise AgxError(f"{what}: no native backward kernel covers this configuration; the ATen backward bridge is "
                       "disabled by default (set AGX_ALLOW_ATEN_BRIDGE=1 to differentiate the ATen restatement instead)")


class _HipForwardAtenBackward(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hip_fn: Callable, aten_fn: Callable, x: Tensor, *params: Tensor):
        ctx.aten_fn = aten_fn
        ctx.params = params
        ctx.save_for_backward(x)
        with torch.no_grad():
            return hip_fn(x.detach())

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        (x,) = ctx.saved_tensors
        params = [p for p in ctx.params if p.requires_grad]
        with torch.enable_grad():
            xl = x.detach().requires_grad_(True)
            out = ctx.aten_fn(xl)
            grads = torch.autograd.grad(out, [xl] + params, grad_out.contiguous(), allow_unused=True)
        it = iter(grads[1:])
        pg = [next(it) if p.requires_grad else None for p in ctx.params]
        return (None, None, grads[0], *pg)


def hip_forward_aten_backward(hip_fn: Callable, aten_fn: Callable, x: Tensor, params: Sequence[Tensor]) -> Tensor:
    """``hip_fn(x)`` with a backward defined by differentiating ``aten_fn(x)``.
    ``params`` are the parameters ``aten_fn`` reads (so that their gradients flow)."""
    require_allowed(getattr(hip_fn, "__qualname__", "hip_forward_aten_backward"))
    return _HipForwardAtenBackward.apply(hip_fn, aten_fn, x, *params)


def needs_grad(x: Tensor, module: torch.nn.Module) -> bool:
    return torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in module.parameters()))
