"""The oracle's blind spot: ``fast_search`` prunes candidates with the same rule as the HIP
kernel, so it must itself be pinned to the FULL defining search (C) and to the independent numpy statement of
the definition -- on the adversarial cases where a wrong bound would prune the true arg-min."""
import numpy as np
import pytest
import torch

from oracle import rvq


def _cases():
    rng = np.random.default_rng(0)
    out = {}
    f = rng.standard_normal((40, 33)).astype(np.float32)
    c = rng.standard_normal((100, 33)).astype(np.float32)
    out["odd_D_plain"] = (f, c)
    c2 = c.copy(); c2[17] = c2[3]; c2[50] = c2[3]; c2[99] = c2[98]
    out["duplicates"] = (f, c2)
    c3 = c.copy(); c3[41] = f[7]; c3[5] = f[7]; c3[60] = f[0]
    out["exact_hits"] = (f, c3)
    off = (3.0 + 100.0 * rng.standard_normal(33)).astype(np.float32)
    out["large_common_offset"] = ((0.02 * f + off).astype(np.float32), (0.02 * c + off).astype(np.float32))
    out["tiny_scale"] = ((1e-6 * f).astype(np.float32), (1e-6 * c).astype(np.float32))
    out["huge_scale"] = ((1e6 * f).astype(np.float32), (1e6 * c).astype(np.float32))
    near = np.repeat(f[:1], 64, axis=0) + 1e-7 * rng.standard_normal((64, 33)).astype(np.float32)
    out["crowded_near_ties"] = (f[:8], near.astype(np.float32))
    out["all_identical"] = (f[:8], np.repeat(c[:1], 32, axis=0))
    out["single_codeword"] = (f[:5], c[:1])
    out["outlier_codeword"] = (f, np.concatenate([c, 1e4 * np.ones((1, 33), np.float32)]))
    fw = rng.standard_normal((12, 512)).astype(np.float32); cw = rng.standard_normal((256, 512)).astype(np.float32)
    out["model_width"] = (fw, cw)
    return out


CASES = _cases()


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("score_dtype", [torch.float64, torch.float32])
def test_fast_search_equals_the_definition(name, score_dtype):
    frames, cb = CASES[name]
    want = rvq.exact_search(frames, cb)
    assert np.array_equal(want, rvq.exact_search_numpy(frames, cb)), "C and numpy statements of the definition differ"
    got = rvq.fast_search(frames, cb, score_dtype)
    assert np.array_equal(got, want), (name, int((got != want).sum()))


def test_residual_quantize_fast_equals_exact_over_stages():
    torch.manual_seed(3)
    x = torch.randn(2, 25, 48) + 1.5
    cbs = torch.randn(4, 96, 48) * torch.tensor([1.0, 0.6, 0.4, 0.25]).view(4, 1, 1)
    cbs[0] += 1.5
    cbs[2, 9] = cbs[2, 2]
    for dt in (torch.float64, torch.float32):
        a = rvq.residual_quantize(x, cbs, method="fast", score_dtype=dt)
        b = rvq.residual_quantize(x, cbs, method="exact")
        assert torch.equal(a[1], b[1]) and torch.equal(a[0], b[0]) and float(a[2]) == float(b[2])
    sizes = (96, 40, 7, 96)
    a = rvq.residual_quantize(x, cbs, method="fast", sizes=sizes)
    b = rvq.residual_quantize(x, cbs, method="exact", sizes=sizes)
    assert torch.equal(a[1], b[1]) and int(a[1][..., 2].max()) < 7
