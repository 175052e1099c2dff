"""Drop-in ``Alibi`` / ``Attention`` / ``FeedForward`` / ``Transformer`` executed
by libagx, plus the bottleneck adapter the reference never shipped.

Mirrors ``networks/transformers.py:7-279`` (class names, constructor arguments,
``state_dict()`` keys ``layers.{i}.0.norm.*``, ``layers.{i}.0.W_{q,k,v,o}.weight``,
``layers.{i}.1.net.{0,1,4}.*``).  Only the branch the reference can actually
execute is implemented -- self-attention with ALiBi (the learned
pos-emb and cross-attention branches raise in the reference); ``depth > 1`` is
build-defined as "every layer uses ALiBi".

Execution is channel-major: the block works on ``(B, C, T)`` tensors (what the
encoder emits), every ``Linear`` is a k=1 convolution on the fp32 MFMA conv
kernel with the activation / residual fused into its epilogue, LayerNorm and
softmax(QK^T + ALiBi)V are the two dedicated kernels of ``csrc/attention.hip``.
The ``nn.LayerNorm`` / ``nn.Linear`` children only hold parameters.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F
from torch import nn

from . import ops
from .autograd_bridge import hip_forward_aten_backward, needs_grad
from ._lib import CONV_CAUSAL, EPI_GELU_PRE, EPI_RESIDUAL, AgxError

Tensor = torch.Tensor


class Alibi(nn.Module):
    """transformers.py:7-93.  ``M`` is a registered (non-persistent) buffer here,
    so it follows ``.to(device)`` -- the reference leaves it on the CPU.
    The attention kernel never reads ``M``: it evaluates ``-slope_h * |i - j|`` itself."""

    def __init__(self, context_x, context_y=None, n_heads=8):
        super().__init__()
        if context_y is not None and context_y != context_x:
            raise NotImplementedError("cross-attention ALiBi (context_y != context_x) has no HIP kernel")
        self.context_x = context_x
        self.context_y = context_x if context_y is None else context_y
        self.n_heads = n_heads
        n_sequence = torch.arange(start=n_heads, end=0, step=-1)
        self.register_buffer("head_scalars", 2 ** (-8 / n_sequence), persistent=False)  # :38-39
        idx = torch.arange(context_x, dtype=torch.float32)
        m = -(idx[:, None] - idx[None, :]).abs()
        self.register_buffer("M", m[None, :] * self.head_scalars[:, None, None], persistent=False)
        self.requires_grad_(False)

    def get_M(self, crop=None):
        m = self.M
        if crop is not None:
            if isinstance(crop, int):
                crop = (crop, crop)
            m = m[:, :crop[0], :crop[1]]
        return m.unsqueeze(0)


class _PackedLinear:
    """Cache of the packed image of one or more ``nn.Linear`` weights stacked
    along the output dim (k=1 conv weights)."""

    def __init__(self):
        self.key = None
        self.packed = None
        self.bias = None

    def get(self, linears):
        key = tuple((l.weight.data_ptr(), l.weight._version,
                     None if l.bias is None else (l.bias.data_ptr(), l.bias._version)) for l in linears)
        if key != self.key:
            w = torch.cat([l.weight.detach() for l in linears], dim=0)
            c_out, c_in = w.shape
            desc = ops.conv_desc(CONV_CAUSAL, 1, c_in, c_out, 1 << 20, 1)
            self.packed = ops.conv_pack(desc, w.reshape(c_out, c_in, 1).contiguous())
            if any(l.bias is not None for l in linears):
                self.bias = torch.cat([l.bias.detach() if l.bias is not None
                                       else torch.zeros(l.out_features, device=w.device) for l in linears])
            else:
                self.bias = None
            self.key = key
        return self.packed, self.bias


def _linear_ct(x: Tensor, packed: Tensor, bias: Optional[Tensor], c_out: int, epilogue: int = 0,
               res: Optional[Tensor] = None) -> Tensor:
    """``Linear`` over the channel dim of a (B, C, T) tensor = k=1 conv."""
    b, c_in, t = x.shape
    desc = ops.conv_desc(CONV_CAUSAL, b, c_in, c_out, t, 1, 1, 1, epilogue)
    return ops.conv_forward(desc, x, packed, bias, res)


class Attention(nn.Module):
    """transformers.py:95-191 (pre-LN multi-head self-attention with ALiBi)."""

    def __init__(self, dim, dim_head=64, n_heads=8, dropout=0., bias=False, context_x=32, context_y=None,
                 has_pos_emb=True, alibi=True):
        super().__init__()
        if not alibi:
            raise NotImplementedError("only the ALiBi branch is defined in the reference (SURVEY 5.1)")
        if context_y is not None:
            raise NotImplementedError("cross-attention has no HIP kernel")
        if dropout != 0.:
            raise NotImplementedError("dropout > 0 is training-only and not on the forward path")
        self.dim, self.dim_head, self.n_heads = dim, dim_head, n_heads
        self.inner_dim = dim_head * n_heads
        self.norm = nn.LayerNorm(dim)
        self.W_q = nn.Linear(dim, self.inner_dim, bias=bias)
        self.W_k = nn.Linear(dim, self.inner_dim, bias=bias)
        self.W_v = nn.Linear(dim, self.inner_dim, bias=bias)
        self.W_o = nn.Linear(self.inner_dim, dim, bias=bias)
        self.dropout = nn.Dropout(dropout)
        self.alibi = alibi
        self.has_pos_emb = has_pos_emb
        self.cross_attention = False
        self.context = context_x
        self.alibi_obj = Alibi(context_x, None, n_heads=n_heads)
        self._qkv, self._o = _PackedLinear(), _PackedLinear()

    def run_bct(self, x: Tensor, residual: Optional[Tensor] = None) -> Tensor:
        """(B, dim, T) -> W_o(attn(LN(x))) [+ residual], channel-major."""
        if x.shape[-1] > self.context:
            raise AgxError(f"sequence length {x.shape[-1]} exceeds the ALiBi context {self.context} "
                           "(the reference fails here too, transformers.py:88-93)")
        xn = ops.layernorm_ct(x, self.norm.weight.detach(), self.norm.bias.detach(), self.norm.eps)
        wqkv, bqkv = self._qkv.get([self.W_q, self.W_k, self.W_v])
        qkv = _linear_ct(xn, wqkv, bqkv, 3 * self.inner_dim)
        o = ops.attention_alibi(qkv, self.alibi_obj.head_scalars, self.n_heads, self.dim_head,
                                self.dim_head ** 0.5)
        wo, bo = self._o.get([self.W_o])
        return _linear_ct(o, wo, bo, self.dim, EPI_RESIDUAL if residual is not None else 0, residual)

    def forward(self, x: Tensor, y=None) -> Tensor:
        """Reference layout: (B, T, dim) -> (B, T, dim)."""
        if y is not None:
            raise NotImplementedError("cross-attention has no HIP kernel")
        return self.run_bct(x.transpose(1, 2).contiguous()).transpose(1, 2).contiguous()


class FeedForward(nn.Module):
    """transformers.py:193-223: LN -> Linear -> exact GELU -> Linear."""

    def __init__(self, dim, hidden_dim, dropout=0., activation=nn.GELU):
        super().__init__()
        if activation is not nn.GELU:
            raise NotImplementedError("only GELU is fused into the FFN kernel epilogue")
        if dropout != 0.:
            raise NotImplementedError("dropout > 0 is training-only and not on the forward path")
        self.net = nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, hidden_dim), activation(), nn.Dropout(dropout),
                                 nn.Linear(hidden_dim, dim), nn.Dropout(dropout))
        self._l1, self._l2 = _PackedLinear(), _PackedLinear()

    def run_bct(self, x: Tensor, residual: Optional[Tensor] = None) -> Tensor:
        ln, l1, l2 = self.net[0], self.net[1], self.net[4]
        xn = ops.layernorm_ct(x, ln.weight.detach(), ln.bias.detach(), ln.eps)
        w1, b1 = self._l1.get([l1])
        h = _linear_ct(xn, w1, b1, l1.out_features, EPI_GELU_PRE)
        w2, b2 = self._l2.get([l2])
        return _linear_ct(h, w2, b2, l2.out_features, EPI_RESIDUAL if residual is not None else 0, residual)

    def forward(self, x: Tensor) -> Tensor:
        return self.run_bct(x.transpose(1, 2).contiguous()).transpose(1, 2).contiguous()


class Transformer(nn.Module):
    """transformers.py:225-279: ``x += attn(x); x += ff(x)`` per layer."""

    def __init__(self, dim, depth=1, heads=8, head_dim=64, dropout=0., context_x=32, context_y=None,
                 has_pos_emb=True, alibi=True):
        super().__init__()
        if context_y is not None:
            raise NotImplementedError("cross-attention has no HIP kernel")
        self.cross_attention = False
        self.layers = nn.ModuleList([
            nn.ModuleList([Attention(dim, n_heads=heads, dim_head=head_dim, dropout=dropout, context_x=context_x,
                                     has_pos_emb=has_pos_emb, alibi=alibi),
                           FeedForward(dim, dim, dropout=dropout)])
            for _ in range(depth)])

    def _hip_bct(self, x: Tensor) -> Tensor:
        for attention, ff in self.layers:
            x = attention.run_bct(x, residual=x)
            x = ff.run_bct(x, residual=x)
        return x

    def _aten_bct(self, x: Tensor) -> Tensor:
        """ATen restatement for the interim backward bridge only (autograd_bridge.py)."""
        x = x.transpose(1, 2)
        for attention, ff in self.layers:
            b, t, _ = x.shape
            xn = attention.norm(x)
            q, k, v = (lin(xn).reshape(b, t, attention.n_heads, attention.dim_head).transpose(1, 2)
                       for lin in (attention.W_q, attention.W_k, attention.W_v))
            s = q @ k.transpose(-1, -2) / (attention.dim_head ** 0.5) + attention.alibi_obj.get_M(crop=(t, t))
            o = (s.softmax(dim=-1) @ v).transpose(1, 2).reshape(b, t, attention.inner_dim)
            x = x + attention.W_o(o)
            x = x + ff.net(x)
        return x.transpose(1, 2)

    def run_bct(self, x: Tensor) -> Tensor:
        """Channel-major (B, dim, T) in and out: 7 launches per layer, both residual adds fused into
        the W_o / FFN-out conv epilogues.  With autograd on, the backward is bridged through ATen."""
        if needs_grad(x, self):
            return hip_forward_aten_backward(self._hip_bct, self._aten_bct, x, list(self.parameters()))
        return self._hip_bct(x)

    def forward(self, x: Tensor, y=None) -> Tensor:
        if y is not None:
            raise NotImplementedError("cross-attention has no HIP kernel")
        return self.run_bct(x.transpose(1, 2).contiguous()).transpose(1, 2).contiguous()


class TransformerBottleneck(nn.Module):
    """Adapter that lets a ``Transformer`` stand where the quantiser does
    (``CausalVQAE.replace_quantizer``, vae.py:347-348; ``Trainer.train_new_quantizer``,
    training.py:502-523).  Honours the quantiser call contract of vae.py:315-318:
    ``(x[b l c], codebook_n, update_codebook=, prioritize_early=) -> (x_out, index, loss)``
    with ``index = None`` and a zero loss (there is nothing to commit to).  The
    reference ships no such adapter; this one is build-defined."""

    def __init__(self, transformer: Transformer, num_quantizers: int = 1):
        super().__init__()
        self.transformer = transformer
        self.num_quantizers = num_quantizers   # training.py:183 reads it
        self.use_som = False                    # utils.py:239

    def quantize_bcl(self, x: Tensor, codebook_n=None, update_codebook=False, prioritize_early=False):
        y = self.transformer.run_bct(x)
        return y, None, torch.zeros((), dtype=torch.float32, device=x.device)

    def forward(self, x: Tensor, codebook_n=None, update_codebook=False, prioritize_early=False):
        y, idx, loss = self.quantize_bcl(x.transpose(1, 2).contiguous())
        return y.transpose(1, 2).contiguous(), idx, loss

    def get_stale_clusters(self):
        return []

    def update_cutoff(self, new_cutoff=None, ratio=None):
        return None
