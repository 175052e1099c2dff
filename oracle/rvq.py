"""Oracle: residual vector quantiser (CPU).

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.  **Parity unpinned**: the
reference's quantiser lives in an external module that is absent from the
reference tree (``networks/vae.py:6``); see the header of ``rvq_exact.c`` for
the algorithm restated here and the arithmetic that defines "bit-exact".

Call-site contract honoured (vae.py:315-318): input ``(B,T,D)`` "b l c",
``codebook_n`` truncates the stage loop, return order ``(x_q, index, commit)``
with ``index`` int64 ``(B,T,Q_used)`` (utils.py:249).
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Tuple

import numpy as np
import torch

from . import build as _build

Tensor = torch.Tensor
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = _build.LIB if os.path.exists(_build.LIB) else _build.build()
        lib = ctypes.CDLL(path)
        fp, ip, lp = (ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int32),
                      ctypes.POINTER(ctypes.c_int64))
        lib.rvq_exact_search.argtypes = [fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, lp]
        lib.rvq_exact_among.argtypes = [fp, fp, ctypes.c_int, ctypes.c_int, ip, ip, ctypes.c_int, lp]
        lib.rvq_apply.argtypes = [fp, fp, fp, lp, ctypes.c_int, ctypes.c_int]
        for f in (lib.rvq_exact_search, lib.rvq_exact_among, lib.rvq_apply):
            f.restype = None
        _LIB = lib
    return _LIB


def _p(a: np.ndarray, ctype):
    return a.ctypes.data_as(ctypes.POINTER(ctype))


def exact_search(frames: np.ndarray, codebook: np.ndarray) -> np.ndarray:
    """Full exact search -- the definition.  frames (N,D) f32, codebook (K,D) f32."""
    frames = np.ascontiguousarray(frames, np.float32)
    codebook = np.ascontiguousarray(codebook, np.float32)
    idx = np.empty(frames.shape[0], np.int64)
    _lib().rvq_exact_search(_p(frames, ctypes.c_float), _p(codebook, ctypes.c_float),
                            frames.shape[0], codebook.shape[0], frames.shape[1],
                            _p(idx, ctypes.c_int64))
    return idx


def exact_search_numpy(frames: np.ndarray, codebook: np.ndarray) -> np.ndarray:
    """The same definition in numpy (sequential over d, no FMA) -- a second,
    independent statement used to cross-check the C on small cases."""
    r = frames.astype(np.float64)[:, None, :]
    c = codebook.astype(np.float64)[None, :, :]
    acc = np.zeros((frames.shape[0], codebook.shape[0]), np.float64)
    for d in range(frames.shape[1]):
        diff = r[:, :, d] - c[:, :, d]
        acc = acc + diff * diff
    return acc.argmin(axis=1).astype(np.int64)  # argmin returns the first minimum


def fast_search(frames: np.ndarray, codebook: np.ndarray, score_dtype=torch.float64) -> np.ndarray:
    """Same result as ``exact_search`` by construction, in O(matmul): scores
    ``|c|^2 - 2 r.c`` from a (float64 by default) matmul pick a candidate set
    with a margin far above the matmul's rounding error; the defining arithmetic
    then decides among the candidates."""
    frames = np.ascontiguousarray(frames, np.float32)
    codebook = np.ascontiguousarray(codebook, np.float32)
    n, dim = frames.shape
    r = torch.from_numpy(frames).to(score_dtype)
    c = torch.from_numpy(codebook).to(score_dtype)
    # distances are translation invariant: centre on the mean codeword so the rounding error of
    # the scores scales with the spread of the data, not with its offset from the origin
    mu = c.mean(dim=0, keepdim=True)
    r, c = r - mu, c - mu
    c2 = (c * c).sum(dim=1)
    scores = c2.unsqueeze(0) - 2.0 * (r @ c.t())
    eps = 1e-12 if score_dtype == torch.float64 else 6e-8
    # per-codeword error bound E[n,k] = (dim+8) eps (|r'_n| + |c'_k|)^2 ; candidate iff its lower
    # bound does not exceed the smallest upper bound (same rule as the HIP kernel)
    rn = (r * r).sum(dim=1).sqrt()
    err = 1.3 * (dim + 8) * eps * (rn.unsqueeze(1) + c2.sqrt().unsqueeze(0)) ** 2
    upper = (scores + err).min(dim=1).values
    mask = (scores - err) <= upper.unsqueeze(1)
    n_cand = mask.sum(dim=1).to(torch.int32)
    max_cand = int(n_cand.max())
    # first max_cand True positions per row (stable order = ascending k)
    order = torch.argsort((~mask).to(torch.int8), dim=1, stable=True)[:, :max_cand]
    cand = np.ascontiguousarray(order.to(torch.int32).numpy())
    n_cand_np = np.ascontiguousarray(n_cand.numpy())
    idx = np.empty(n, np.int64)
    _lib().rvq_exact_among(_p(frames, ctypes.c_float), _p(codebook, ctypes.c_float), n, dim,
                           _p(cand, ctypes.c_int32), _p(n_cand_np, ctypes.c_int32), max_cand,
                           _p(idx, ctypes.c_int64))
    return idx


def residual_quantize(x: Tensor, codebooks: Tensor, codebook_n: Optional[int] = None,
                      method: str = "fast", score_dtype=torch.float64, sizes=None
                      ) -> Tuple[Tensor, Tensor, Tensor]:
    """RVQ forward in eval mode.

    x (B,T,D) f32, codebooks (Q,K,D) f32 ->
      x_q (B,T,D) f32 = sum of the selected codewords in stage order,
      index (B,T,Q_used) int64,
      commit loss (scalar f32) = sum over stages of mean((r_in - c_sel)^2),
        accumulated in float64 (the GPU side is compared with a tolerance).
    ``sizes[q]`` (optional): stage q searches only its first sizes[q] codewords -- one codebook size per
    quantizer (the reference's ``codebook_size`` tuple, vae.py:233) stored in a (Q, max K, D) tensor.
    """
    q_total, k, d = codebooks.shape
    q_used = q_total if codebook_n is None else max(0, min(int(codebook_n), q_total))
    b, t, _ = x.shape
    frames = np.ascontiguousarray(x.detach().reshape(b * t, d).to(torch.float32).numpy()).copy()
    out = np.zeros_like(frames)
    index = np.empty((b * t, q_used), np.int64)
    commit = 0.0
    cbs = codebooks.detach().to(torch.float32).contiguous().numpy()
    for q in range(q_used):
        cb = np.ascontiguousarray(cbs[q] if sizes is None else cbs[q][:int(sizes[q])])
        if method == "exact":
            idx = exact_search(frames, cb)
        else:
            idx = fast_search(frames, cb, score_dtype)
        _lib().rvq_apply(_p(frames, ctypes.c_float), _p(out, ctypes.c_float),
                         _p(cb, ctypes.c_float), _p(idx, ctypes.c_int64), b * t, d)
        index[:, q] = idx
        commit += float(np.mean(frames.astype(np.float64) ** 2))
    return (torch.from_numpy(out).reshape(b, t, d), torch.from_numpy(index).reshape(b, t, q_used),
            torch.tensor(commit, dtype=torch.float32))


def residual_quantize_train(x: Tensor, codebooks: Tensor, codebook_n: Optional[int] = None
                            ) -> Tuple[Tensor, Tensor, Tensor]:
    """Training-mode semantics of the build's quantiser (build-defined; the external module's
    are unknown): straight-through estimator ``x + (x_q - x).detach()`` and the commitment
    loss ``sum_q mean((x - sum_{p<=q} c_p)^2)`` with the selected codewords detached, so the
    loss is differentiable in the encoder output.  Differentiable torch ops on CPU."""
    xq, index, _ = residual_quantize(x.detach(), codebooks, codebook_n)
    partial, commit = None, x.new_zeros(())
    for q in range(index.shape[-1]):
        c = codebooks.detach()[q][index[..., q]]
        partial = c if partial is None else partial + c
        commit = commit + ((x - partial) ** 2).mean()
    return x + (xq - x).detach(), index, commit


def dequantize(codebook: Tensor, idx: Tensor) -> Tensor:
    """``quantizers[i].dequantize(idx)`` (call site vae.py:333): plain gather,
    (..,) int -> (.., D)."""
    return codebook[idx]


def init_codebooks(q: int, k: int, d: int, sigma: float = 1.0, seed: int = 7) -> Tensor:
    """``randn(Q,K,D) * sigma`` from seed 7."""
    gen = torch.Generator().manual_seed(seed)
    return torch.randn(q, k, d, generator=gen) * sigma


def ema_assignment_stats(frames: torch.Tensor, codebooks: torch.Tensor, index: torch.Tensor) -> torch.Tensor:
    """CPU statement of ``agx_rvq_ema_stats`` (build-defined EMA codebook update; the external quantiser's own rule is
    unknown): frames (N, D), codebooks (Q, K, D) BEFORE the update, index (N, q_used) ->
    stats (q_used, K, D + 1) with [q, k, 0] = assignment count and [q, k, 1:] = sum of the stage-q residuals of the
    frames assigned to k -- the residual the search saw, i.e. against the pre-update codewords."""
    n, d = frames.shape
    q_used, k = index.shape[1], codebooks.shape[1]
    stats = torch.zeros(q_used, k, d + 1, dtype=frames.dtype)
    residual = frames.clone()
    for q in range(q_used):
        idx = index[:, q]
        stats[q, :, 0] = torch.bincount(idx, minlength=k).to(frames.dtype)
        stats[q, :, 1:].index_add_(0, idx, residual)
        residual = residual - codebooks[q][idx]
    return stats
