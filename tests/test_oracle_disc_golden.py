"""The discriminator oracle against the reference's own outputs (tests/golden/g7_discriminators.npz,
written by tests/golden/make_goldens_disc.py from networks/discriminator.py)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import discriminator as od

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def g7():
    z = np.load(os.path.join(GOLD, "g7_discriminators.npz"))
    return {k: torch.from_numpy(z[k]) for k in z.files}


@pytest.fixture(scope="module")
def meta():
    with open(os.path.join(GOLD, "meta_g7.json")) as f:
        return json.load(f)


def sub(g, prefix):
    return {k[len(prefix):]: v.clone() for k, v in g.items() if k.startswith(prefix)}


def close(a, b, tol=2e-5):
    scale = float(b.abs().max()) + 1e-12
    assert a.shape == b.shape
    assert float((a - b).abs().max()) <= tol * scale + 1e-7, float((a - b).abs().max()) / scale


@pytest.mark.parametrize("win", [64, 256, 1024])
def test_stft_front_end(g7, win):
    x, want = g7[f"stft_only_{win}/x"], g7[f"stft_only_{win}/y"]          # want (B, F, T, 2)
    got = od.stft_two_sided(x, win, win // 4)                               # (B, 2, T, F)
    close(got, want.permute(0, 3, 2, 1), 1e-5)


@pytest.mark.parametrize("scale", [1, 2])
@pytest.mark.parametrize("train", [False, True])
def test_waveform_block(g7, meta, scale, train):
    name = f"wave_s{scale}_{'train' if train else 'eval'}"
    sd = sub(g7, name + "/sd/")
    groups = meta["wave"]["kwargs"]["groups"]
    out, feats = od.waveform_block(g7[name + "/x"], sd, "", scale, train, groups=groups)
    close(out, g7[name + "/out0"])
    if not train:
        assert len(feats) == meta["wave"]["n_features"]
        for i, f in enumerate(feats):
            close(f, g7[f"{name}/feat{i}"])
    else:
        moved = 0
        for k, v in sub(g7, name + "/sd_after/").items():
            close(sd[k], v, 1e-5)
            moved += not torch.equal(sd[k], g7[f"{name}/sd/{k}"])
        assert moved >= 12                                               # the power iteration ran


@pytest.mark.parametrize("train", [False, True])
def test_stft_discriminator(g7, meta, train):
    name = f"stft_{'train' if train else 'eval'}"
    sd = sub(g7, name + "/sd/")
    outs, feats = od.stft_discriminator(g7[name + "/x"], sd, meta["stft"]["kwargs"]["win_length"], train)
    close(outs[0], g7[name + "/out0"])
    if not train:
        assert len(feats) == meta["stft"]["n_features"]
        for i, f in enumerate(feats):
            close(f, g7[f"{name}/feat{i}"], 5e-5)
    else:
        for k, v in sub(g7, name + "/sd_after/").items():
            close(sd[k], v, 1e-5)


@pytest.mark.parametrize("tag", ["wave", "stft"])
def test_loss(g7, meta, tag):
    sd = sub(g7, f"loss_{tag}/sd/")
    orig, rec = g7["loss/original"], g7["loss/reconstruction"]
    if tag == "wave":
        groups = meta["wave"]["kwargs"]["groups"]
        disc = lambda t: od.waveform_discriminator(t, sd, n_blocks=2, train=True, groups=groups)  # noqa: E731
    else:
        disc = lambda t: od.stft_discriminator(t, sd, meta["stft"]["kwargs"]["win_length"], train=True)  # noqa: E731
    gl, dl = od.discriminator_generator_loss(orig, rec, disc)
    close(gl, g7[f"loss_{tag}/generator_loss"], 1e-4)
    close(dl, g7[f"loss_{tag}/discriminator_loss"], 1e-5)
    gl2, dl2 = od.discriminator_generator_loss(orig, rec, disc, feature_multiplier=3.0, scale_feature_loss=False)
    close(gl2, g7[f"loss_{tag}/generator_loss_unscaled_fm3"], 1e-4)
    close(dl2, g7[f"loss_{tag}/discriminator_loss_2nd_call"], 1e-5)
