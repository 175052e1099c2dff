"""Oracle: proof obligation for index disagreements between two RVQ runs on slightly different latents.

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.  Used by ``tests/`` and by ``bench.py``'s parity block.

Setting (hard part 2; call site ``networks/vae.py:315-318``).  Path A (the HIP encoder) and path B
(the CPU oracle's encoder) produce latents that differ by fp32 rounding; each path's search is bit-exact against the
definition (``rvq_exact.c``) on ITS OWN latents.  Where the two index tensors differ, that must be explained by the
latent difference alone -- a search bug must not hide behind "agreement > 0.97".

For a frame whose first differing stage is q*, both paths subtracted the same codewords before q*, so the stage
residuals are ``r_a`` and ``r_b = r_a - delta`` with ``delta`` = the latent difference (up to the binary32 rounding
of the subtractions, which is carried along: the residuals are re-formed in binary32 exactly as ``rvq_apply`` does).
With ``d(r, k) = |r - c_k|^2`` (binary64), path A chose ``k_a``, path B chose ``k_b``:

    m_a = d(r_a, k_b) - d(r_a, k_a) >= 0        (k_a is the arg-min on r_a)
    m_b = d(r_b, k_a) - d(r_b, k_b) >= 0        (k_b is the arg-min on r_b)
    m_a + m_b = 2 <delta, c_ka - c_kb>  <=  2 |delta| |c_ka - c_kb|  =: eps

so BOTH top-2 margins are bounded by ``eps``, a number derived from the MEASURED latent error of that frame.  A
disagreement with ``m_a > eps`` or ``m_b > eps`` (or a negative margin) cannot come from the latent difference: it is
a search bug.  Stages after q* are not comparable (the residuals have genuinely diverged) and are only counted.
"""
from __future__ import annotations

from typing import Dict

import numpy as np


def _dist64(r: np.ndarray, c: np.ndarray) -> np.ndarray:
    d = r.astype(np.float64) - c.astype(np.float64)
    return (d * d).sum(axis=-1)


def explain_disagreements(z_a, z_b, idx_a, idx_b, codebooks, slack: float = 1e-9) -> Dict[str, float]:
    """z_a, z_b: (N, D) float32 latent frames of the two paths; idx_a, idx_b: (N, Q) int; codebooks (Q, K, D) f32.

    Returns counts and the worst ratios; ``proved`` is True iff every first disagreement is a near tie within the
    bound derived above.  ``slack`` (relative to the distance itself) covers the binary64 evaluation of the margins."""
    z_a = np.ascontiguousarray(np.asarray(z_a, dtype=np.float32))
    z_b = np.ascontiguousarray(np.asarray(z_b, dtype=np.float32))
    idx_a = np.asarray(idx_a).astype(np.int64)
    idx_b = np.asarray(idx_b).astype(np.int64)
    cbs = np.asarray(codebooks, dtype=np.float32)
    n, q_used = idx_a.shape
    assert z_a.shape == z_b.shape == (n, cbs.shape[2]) and idx_b.shape == idx_a.shape

    differ = idx_a != idx_b
    frames = np.nonzero(differ.any(axis=1))[0]
    first = differ[frames].argmax(axis=1)                 # first differing stage of each such frame
    out = {
        "frames": int(n), "stages": int(q_used),
        "frames_with_a_disagreement": int(frames.size),
        "codes_disagreeing": int(differ.sum()),
        "agreement": float(1.0 - differ.mean()),
        "first_disagreement_by_stage": [int((first == q).sum()) for q in range(q_used)],
        "max_margin_over_bound": 0.0, "max_relative_margin": 0.0, "max_bound_relative": 0.0,
        "max_latent_error_norm": float(np.sqrt(((z_a.astype(np.float64) - z_b) ** 2).sum(axis=1)).max()) if n else 0.0,
        "max_latent_error_relative": 0.0, "negative_margins": 0, "unexplained": 0,
    }
    if n:
        zn = np.sqrt((z_b.astype(np.float64) ** 2).sum(axis=1))
        dn = np.sqrt(((z_a.astype(np.float64) - z_b) ** 2).sum(axis=1))
        out["max_latent_error_relative"] = float((dn / np.maximum(zn, 1e-30)).max())
    for f, qs in zip(frames, first):
        ra, rb = z_a[f].copy(), z_b[f].copy()
        for p in range(qs):                                # binary32 residual updates, stage order (rvq_apply)
            c = cbs[p, idx_a[f, p]]
            ra = ra - c
            rb = rb - c
        ca, cb = cbs[qs, idx_a[f, qs]], cbs[qs, idx_b[f, qs]]
        m_a = _dist64(ra, cb) - _dist64(ra, ca)
        m_b = _dist64(rb, ca) - _dist64(rb, cb)
        delta = ra.astype(np.float64) - rb.astype(np.float64)
        eps = 2.0 * np.sqrt((delta * delta).sum()) * np.sqrt(_dist64(ca, cb))
        scale = max(float(_dist64(ra, ca)), float(_dist64(rb, cb)), 1e-300)
        tol = slack * scale
        if m_a < -tol or m_b < -tol:
            out["negative_margins"] += 1
        if max(m_a, m_b) > eps + tol:
            out["unexplained"] += 1
        out["max_margin_over_bound"] = max(out["max_margin_over_bound"], float(max(m_a, m_b) / max(eps, 1e-300)))
        out["max_relative_margin"] = max(out["max_relative_margin"], float(max(m_a, m_b) / scale))
        out["max_bound_relative"] = max(out["max_bound_relative"], float(eps / scale))
    out["proved"] = bool(out["negative_margins"] == 0 and out["unexplained"] == 0)
    return out
