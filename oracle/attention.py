"""Oracle: ALiBi multi-head attention block + feed-forward (CPU, torch fp32).

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.
Follows ``networks/transformers.py``: ``Alibi`` (:7-93), ``Attention.forward``
(:157-191, self-attention + ALiBi branch only -- the other branches raise in
the reference), ``FeedForward`` (:213-220), ``Transformer.forward``
(:275-279, depth 1 is the only depth the reference can construct).
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def alibi_slopes(n_heads: int) -> Tensor:
    """transformers.py:38-39: ``2 ** (-8 / n)`` for n = H, H-1, ..., 1
    (head 0 gets the *largest* slope 2^(-8/H); this is the reference's
    formula, not the ALiBi paper's)."""
    n = torch.arange(n_heads, 0, -1, dtype=torch.float32)
    return 2.0 ** (-8.0 / n)


def alibi_bias(n_heads: int, t_q: int, t_k: int) -> Tensor:
    """(H, t_q, t_k) bias ``-slope_h * |i - j|`` -- the square, symmetric
    self-attention case of ``Alibi._create_M`` (:45-77) cropped as ``get_M``
    does (:88-93)."""
    i = torch.arange(t_q, dtype=torch.float32).reshape(-1, 1)
    j = torch.arange(t_k, dtype=torch.float32).reshape(1, -1)
    return -(i - j).abs().unsqueeze(0) * alibi_slopes(n_heads).reshape(-1, 1, 1)


def attention(x: Tensor, sd: Dict[str, Tensor], prefix: str, n_heads: int) -> Tensor:
    """transformers.py:157-191.  ``x`` (B,T,dim) -> (B,T,dim); parameters under
    ``prefix``: ``norm.{weight,bias}``, ``W_{q,k,v,o}.weight`` (bias-free)."""
    b, t, dim = x.shape
    xn = F.layer_norm(x, (dim,), sd[prefix + "norm.weight"], sd[prefix + "norm.bias"])
    q = F.linear(xn, sd[prefix + "W_q.weight"], sd.get(prefix + "W_q.bias"))
    k = F.linear(xn, sd[prefix + "W_k.weight"], sd.get(prefix + "W_k.bias"))
    v = F.linear(xn, sd[prefix + "W_v.weight"], sd.get(prefix + "W_v.bias"))
    dh = q.shape[-1] // n_heads
    q, k, v = (z.reshape(b, t, n_heads, dh).transpose(1, 2) for z in (q, k, v))
    s = q @ k.transpose(-1, -2)
    s = s / (dh ** 0.5)
    s = s + alibi_bias(n_heads, t, t).unsqueeze(0)
    p = s.softmax(dim=-1)
    o = (p @ v).transpose(1, 2).reshape(b, t, n_heads * dh)
    return F.linear(o, sd[prefix + "W_o.weight"], sd.get(prefix + "W_o.bias"))


def feed_forward(x: Tensor, sd: Dict[str, Tensor], prefix: str) -> Tensor:
    """transformers.py:213-220: LN -> Linear -> exact GELU -> Linear
    (``prefix`` + ``net.{0,1,4}``)."""
    dim = x.shape[-1]
    h = F.layer_norm(x, (dim,), sd[prefix + "net.0.weight"], sd[prefix + "net.0.bias"])
    h = F.gelu(F.linear(h, sd[prefix + "net.1.weight"], sd[prefix + "net.1.bias"]))
    return F.linear(h, sd[prefix + "net.4.weight"], sd[prefix + "net.4.bias"])


def transformer(x: Tensor, sd: Dict[str, Tensor], n_heads: int, depth: int = 1,
                prefix: str = "") -> Tensor:
    """transformers.py:275-279.  ``depth > 1`` stacks the same block structure
    (the reference itself cannot build depth > 1; the build defines
    deeper stacks as every layer using ALiBi)."""
    for layer in range(depth):
        p = f"{prefix}layers.{layer}."
        x = x + attention(x, sd, p + "0.", n_heads)
        x = x + feed_forward(x, sd, p + "1.")
    return x


def init_state_dict(dim: int, n_heads: int, head_dim: int, depth: int = 1, seed: int = 0,
                    prefix: str = "") -> Dict[str, Tensor]:
    """Seeded parameters in the reference's ``Transformer.state_dict()`` layout
    (torch default ``Linear`` init family; LayerNorm weight 1 / bias 0 jittered
    so the affine part is exercised)."""
    gen = torch.Generator().manual_seed(seed)
    inner = n_heads * head_dim
    sd: Dict[str, Tensor] = {}

    def uni(shape, fan_in):
        return (torch.rand(shape, generator=gen) * 2 - 1) / (fan_in ** 0.5)

    for layer in range(depth):
        p = f"{prefix}layers.{layer}."
        sd[p + "0.norm.weight"] = 1 + 0.1 * uni((dim,), 1)
        sd[p + "0.norm.bias"] = 0.1 * uni((dim,), 1)
        sd[p + "0.W_q.weight"] = uni((inner, dim), dim)
        sd[p + "0.W_k.weight"] = uni((inner, dim), dim)
        sd[p + "0.W_v.weight"] = uni((inner, dim), dim)
        sd[p + "0.W_o.weight"] = uni((dim, inner), inner)
        sd[p + "1.net.0.weight"] = 1 + 0.1 * uni((dim,), 1)
        sd[p + "1.net.0.bias"] = 0.1 * uni((dim,), 1)
        sd[p + "1.net.1.weight"] = uni((dim, dim), dim)
        sd[p + "1.net.1.bias"] = uni((dim,), dim)
        sd[p + "1.net.4.weight"] = uni((dim, dim), dim)
        sd[p + "1.net.4.bias"] = uni((dim,), dim)
    return sd
