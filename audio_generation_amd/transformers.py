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
from torch import nn

from . import ops
from ._lib import CONV_CAUSAL, EPI_GELU_PRE, EPI_RESIDUAL, AgxError, needs_grad

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
        self.w3d = None

    def get(self, linears):
        key = tuple((l.weight.data_ptr(), l.weight._version,
                     None if l.bias is None else (l.bias.data_ptr(), l.bias._version)) for l in linears)
        if key != self.key:
            w = torch.cat([l.weight.detach() for l in linears], dim=0)
            c_out, c_in = w.shape
            desc = ops.conv_desc(CONV_CAUSAL, 1, c_in, c_out, 1 << 20, 1)
            self.w3d = w.reshape(c_out, c_in, 1).contiguous()      # the native backward reads it
            self.packed = ops.conv_pack(desc, self.w3d)
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
        # arithmetic of the QK^T / PV contractions: "fp32" (exact, the reference's) or "bf16" (bf16 MFMA, fp32 accumulate
        # and softmax -- BASELINE config 3); inference only: with autograd on, the block's forward runs the fp32 kernel
        # (Transformer.run_bct), which is what the backward kernels differentiate
        self.attention_dtype = "fp32"

    def run_bct(self, x: Tensor, residual: Optional[Tensor] = None) -> Tensor:
        """(B, dim, T) -> W_o(attn(LN(x))) [+ residual], channel-major."""
        if x.shape[-1] > self.context:
            raise AgxError(f"sequence length {x.shape[-1]} exceeds the ALiBi context {self.context} "
                           "(the reference fails here too, transformers.py:88-93)")
        xn = ops.layernorm_ct(x, self.norm.weight.detach(), self.norm.bias.detach(), self.norm.eps)
        wqkv, bqkv = self._qkv.get([self.W_q, self.W_k, self.W_v])
        qkv = _linear_ct(xn, wqkv, bqkv, 3 * self.inner_dim)
        o = ops.attention_alibi(qkv, self.alibi_obj.head_scalars, self.n_heads, self.dim_head,
                                self.dim_head ** 0.5,
                                ops.ATTN_BF16 if self.attention_dtype == "bf16" else ops.ATTN_FP32)
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


def _lin_bwd(pl: "_PackedLinear", x_in: Tensor, dy: Tensor, c_out: int, pre: Optional[Tensor] = None,
             need_dx: bool = True):
    """Backward of ``_linear_ct`` (k=1 conv): (dx or None, dW (c_out, c_in), dbias or None); with ``pre`` the
    GELU gradient of the layer BELOW (at its pre-activation) is fused into the bwd-data epilogue."""
    b, c_in, t = x_in.shape
    desc = ops.conv_desc(CONV_CAUSAL, b, c_in, c_out, t, 1)
    dw, _, db = ops.conv_bwd_weight(desc, x_in, dy, pl.w3d, None, want_bias=pl.bias is not None)
    dx = None
    if need_dx:
        pk = ops.conv_pack_bwd(desc, pl.w3d)
        dx = ops.conv_bwd_data(desc, dy, pk) if pre is None else ops.conv_bwd_data_gelu(desc, dy, pk, pre)
    return dx, dw.reshape(c_out, c_in), db


class _TransformerNative(torch.autograd.Function):
    """Transformer forward + hand-written backward on the HIP kernels: k=1 conv backward for every Linear,
    ``agx_attention_alibi_backward``, ``agx_layernorm_ct_backward`` (residual adds fused as ``add``), the GELU
    gradient in a bwd-data epilogue (the pre-activation is recomputed with one conv launch)."""

    @staticmethod
    def forward(ctx, tf, x: Tensor, *params: Tensor):
        saved = []
        with torch.no_grad():
            h = x.detach()
            for attention, ff in tf.layers:
                ln1 = attention.norm
                xn1 = ops.layernorm_ct(h, ln1.weight.detach(), ln1.bias.detach(), ln1.eps)
                wqkv, bqkv = attention._qkv.get([attention.W_q, attention.W_k, attention.W_v])
                qkv = _linear_ct(xn1, wqkv, bqkv, 3 * attention.inner_dim)
                o = ops.attention_alibi(qkv, attention.alibi_obj.head_scalars, attention.n_heads, attention.dim_head,
                                        attention.dim_head ** 0.5)
                wo, bo = attention._o.get([attention.W_o])
                x1 = _linear_ct(o, wo, bo, attention.dim, EPI_RESIDUAL, h)
                ln2, l1, l2 = ff.net[0], ff.net[1], ff.net[4]
                xn2 = ops.layernorm_ct(x1, ln2.weight.detach(), ln2.bias.detach(), ln2.eps)
                w1, b1 = ff._l1.get([l1])
                hid = _linear_ct(xn2, w1, b1, l1.out_features, EPI_GELU_PRE)
                w2, b2 = ff._l2.get([l2])
                x2 = _linear_ct(hid, w2, b2, l2.out_features, EPI_RESIDUAL, x1)
                saved += [h, xn1, qkv, o, x1, xn2, hid]
                h = x2
        ctx.tf, ctx.params = tf, params
        ctx.save_for_backward(*saved)
        return h

    @staticmethod
    def backward(ctx, g: Tensor):
        tf, saved = ctx.tf, ctx.saved_tensors
        g = g.contiguous()
        grads = {}
        for li in range(len(tf.layers) - 1, -1, -1):
            attention, ff = tf.layers[li]
            h, xn1, qkv, o, x1, xn2, hid = saved[7 * li:7 * li + 7]
            ln2, l1, l2 = ff.net[0], ff.net[1], ff.net[4]
            # x2 = x1 + W2 gelu(W1 LN2(x1) + b1) + b2
            w1, b1 = ff._l1.get([l1])
            pre = _linear_ct(xn2, w1, b1, l1.out_features)                 # pre-activation, recomputed
            dpre, dw2, db2 = _lin_bwd(ff._l2, hid, g, l2.out_features, pre=pre)
            dxn2, dw1, db1 = _lin_bwd(ff._l1, xn2, dpre, l1.out_features)
            dx1, dg2, dbt2 = ops.layernorm_ct_backward(x1, ln2.weight.detach(), dxn2, ln2.eps, add=g)
            grads[l2.weight], grads[l1.weight], grads[ln2.weight], grads[ln2.bias] = dw2, dw1, dg2, dbt2
            if l2.bias is not None:
                grads[l2.bias] = db2
            if l1.bias is not None:
                grads[l1.bias] = db1
            # x1 = h + W_o attn(W_qkv LN1(h))
            ln1, inner = attention.norm, attention.inner_dim
            do, dwo, dbo = _lin_bwd(attention._o, o, dx1, attention.dim)
            dqkv = ops.attention_alibi_backward(qkv, attention.alibi_obj.head_scalars, do, attention.n_heads,
                                                attention.dim_head, attention.dim_head ** 0.5, out=o)
            need_dx = li > 0 or ctx.needs_input_grad[1]
            dxn1, dwqkv, dbqkv = _lin_bwd(attention._qkv, xn1, dqkv, 3 * inner)
            dx, dg1, dbt1 = ops.layernorm_ct_backward(h, ln1.weight.detach(), dxn1, ln1.eps, add=dx1)
            grads[attention.W_o.weight], grads[ln1.weight], grads[ln1.bias] = dwo, dg1, dbt1
            for k, lin in enumerate((attention.W_q, attention.W_k, attention.W_v)):
                grads[lin.weight] = dwqkv[k * inner:(k + 1) * inner]
                if lin.bias is not None:
                    grads[lin.bias] = dbqkv[k * inner:(k + 1) * inner]
            if attention.W_o.bias is not None:
                grads[attention.W_o.bias] = dbo
            g = dx
            del need_dx
        return (None, g if ctx.needs_input_grad[1] else None, *[grads.get(p_) for p_ in ctx.params])


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

    def run_bct(self, x: Tensor) -> Tensor:
        """Channel-major (B, dim, T) in and out: 7 launches per layer, both residual adds fused into
        the W_o / FFN-out conv epilogues.  With autograd on, the backward runs on the HIP kernels too
        (_TransformerNative: head_dim <= 128; the training forward always uses the fp32 attention arithmetic, whatever
        ``attention_dtype`` says -- the backward kernels recompute P from fp32 scores)."""
        for attention, _ in self.layers:
            if x.shape[-1] > attention.context:
                raise AgxError(f"sequence length {x.shape[-1]} exceeds the ALiBi context {attention.context} "
                               "(the reference fails here too, transformers.py:88-93)")
        if needs_grad(x, self):
            if any(a.dim_head > 128 for a, _ in self.layers):
                raise AgxError("Transformer: the attention backward kernels cover head_dim <= 128 "
                               "(agx_attention_alibi_backward_ex); larger heads run forward only -- there is no ATen fallback")
            return _TransformerNative.apply(self, x, *list(self.parameters()))
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
