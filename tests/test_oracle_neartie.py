"""The near-tie checker itself (``oracle/neartie.py``): it must accept index flips that come from an fp32-sized latent
difference and reject anything else -- otherwise the GPU parity tests that rely on it prove nothing."""
import numpy as np
import torch

from oracle import neartie, rvq


def _twin_codebooks(frames, q, k, seed, pairs=None):
    """Stage 0 = frames + noise, except ``pairs`` engineered near ties: codewords 2i, 2i+1 = z_f +- u for a chosen frame
    f, so that frame sits on the bisector of the two up to binary32 rounding and a rounding-sized change of the latent
    decides between them; later stages shrinking randn."""
    gen = torch.Generator().manual_seed(seed)
    d = frames.shape[1]
    sigma = float(frames.std())
    cbs = torch.randn(q, k, d, generator=gen) * sigma
    pick = torch.randint(0, frames.shape[0], (k,), generator=gen)
    cbs[0] = frames[pick] + 0.1 * sigma * torch.randn(k, d, generator=gen)
    pairs = k // 4 if pairs is None else pairs
    u = 0.05 * sigma * torch.randn(pairs, d, generator=gen)
    centre = frames[torch.randperm(frames.shape[0], generator=gen)[:pairs]]
    if pairs:
        cbs[0, 0:2 * pairs:2] = centre + u
        cbs[0, 1:2 * pairs:2] = centre - u
    for s in range(1, q):
        cbs[s] *= 0.6 ** s
    return cbs


def test_flips_from_a_rounding_sized_latent_difference_are_proved_near_ties():
    gen = torch.Generator().manual_seed(1)
    n, d, q, k = 600, 64, 3, 128
    z_b = torch.randn(1, n, d, generator=gen) + 1.5
    z_a = z_b + 2e-6 * torch.randn(1, n, d, generator=gen)          # "the other encoder's rounding"
    cbs = _twin_codebooks(z_b[0], q, k, 2)
    _, idx_a, _ = rvq.residual_quantize(z_a, cbs, method="exact")
    _, idx_b, _ = rvq.residual_quantize(z_b, cbs, method="exact")
    rep = neartie.explain_disagreements(z_a[0].numpy(), z_b[0].numpy(), idx_a[0].numpy(), idx_b[0].numpy(), cbs.numpy())
    assert rep["frames_with_a_disagreement"] > 10, rep      # the construction does produce flips ...
    assert rep["proved"] and rep["max_margin_over_bound"] <= 1.0 + 1e-6, rep   # ... and each one is a near tie
    assert rep["max_relative_margin"] < 1e-4, rep


def test_a_search_bug_is_not_explained():
    gen = torch.Generator().manual_seed(3)
    n, d, q, k = 400, 64, 3, 128
    z_b = torch.randn(1, n, d, generator=gen)
    z_a = z_b + 1e-6 * torch.randn(1, n, d, generator=gen)
    cbs = _twin_codebooks(z_b[0], q, k, 4, pairs=0)       # no engineered ties: every flip below is the bug
    _, idx_a, _ = rvq.residual_quantize(z_a, cbs, method="exact")
    _, idx_b, _ = rvq.residual_quantize(z_b, cbs, method="exact")
    bad = idx_a.clone()
    rows = torch.arange(0, n, 50)                                    # 2 % of the frames: the share a bare
    bad[0, rows, 1] = (bad[0, rows, 1] + 17) % k                     # "agreement > 0.97" assert lets through
    rep = neartie.explain_disagreements(z_a[0].numpy(), z_b[0].numpy(), bad[0].numpy(), idx_b[0].numpy(), cbs.numpy())
    assert not rep["proved"] and rep["unexplained"] + rep["negative_margins"] >= rows.numel() - 1, rep
    assert rep["agreement"] > 0.97                                   # ... which the old assert would have passed


def test_identical_runs_have_nothing_to_explain():
    z = np.random.default_rng(0).standard_normal((50, 16)).astype(np.float32)
    cbs = np.random.default_rng(1).standard_normal((2, 8, 16)).astype(np.float32)
    _, idx, _ = rvq.residual_quantize(torch.from_numpy(z)[None], torch.from_numpy(cbs), method="exact")
    rep = neartie.explain_disagreements(z, z, idx[0].numpy(), idx[0].numpy(), cbs)
    assert rep["proved"] and rep["frames_with_a_disagreement"] == 0 and rep["agreement"] == 1.0
