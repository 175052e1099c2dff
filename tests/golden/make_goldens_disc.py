"""Generate ``g7_discriminators.npz`` from the REFERENCE's ``networks/discriminator.py``.

Same rules as ``make_goldens.py`` (run in the build container only; the reference modules are imported
as they are, with the same two empty placeholder modules; only numeric inputs / outputs are written).
Kept separate so that the fixtures G1-G6 are not rewritten.
"""
from __future__ import annotations

import json
import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_goldens import OUT, REF, _install_placeholders  # noqa: E402


def _np(sd):
    # copies: spectral norm updates its u / v buffers in place
    return {k: v.detach().cpu().numpy().copy() for k, v in sd.items()}


def main():
    _install_placeholders()
    sys.path.insert(0, REF)
    import discriminator as ref_d  # noqa: E402  (reference)

    torch.set_num_threads(4)
    g, meta = {}, {}

    def run_disc(name, disc, x, train):
        """state before, outputs + features, state after (spectral-norm u / v move in train mode)."""
        disc.train(train)
        for k, v in _np(disc.state_dict()).items():
            g[f"{name}/sd/{k}"] = v
        with torch.no_grad():
            outs, feats = disc(x)
        if not isinstance(outs, (list, tuple)):
            outs = [outs]
        g[f"{name}/x"] = x.numpy()
        for i, o in enumerate(outs):
            g[f"{name}/out{i}"] = o.numpy().copy()
        if not train:                      # features once (eval); the train-mode run pins out + u / v
            for i, f in enumerate(feats):
                g[f"{name}/feat{i}"] = f.numpy().copy()
        for k, v in _np(disc.state_dict()).items():
            if k.endswith("_u") or k.endswith("_v"):
                g[f"{name}/sd_after/{k}"] = v
        return len(outs), len(feats)

    # ---- waveform discriminator block (reduced widths, the reference's kernel sizes / strides) -----
    torch.manual_seed(7)
    wkw = dict(channel_sizes=[4, 8, 16, 16, 32, 32, 32], groups=[1, 2, 4, 4, 8, 1, 1])
    for scale in (1, 2):
        blk = ref_d.WaveformDiscriminatorBlock(1, scale=scale, **wkw)
        x = 0.3 * torch.randn(1, 1, 8192 * scale)
        for train in (False, True):
            n_out, n_feat = run_disc(f"wave_s{scale}_{'train' if train else 'eval'}", blk, x, train)
    meta["wave"] = {"kwargs": wkw, "scales": [1, 2], "n_features": n_feat}

    # the default block's layer table (shapes only; weights are too large for a fixture)
    full = ref_d.WaveformDiscriminatorBlock(1)
    rows = []
    for name, mod in full.named_modules():
        if isinstance(mod, torch.nn.Conv1d):
            rows.append({"name": name, "cin": mod.in_channels, "cout": mod.out_channels, "k": mod.kernel_size[0],
                         "stride": mod.stride[0], "groups": mod.groups, "padding": mod.padding[0]})
    meta["wave_default_layers"] = rows
    meta["wave_default_keys"] = list(full.state_dict().keys())

    # ---- STFT discriminator (reduced widths / window) --------------------------------------------
    torch.manual_seed(8)
    skw = dict(in_channels=2, first_channel_size=4, win_length=256)
    sd_ = ref_d.STFTDiscriminator(**skw)
    x = 0.3 * torch.randn(1, 1, 1024)
    for train in (False, True):
        n_out, n_feat = run_disc(f"stft_{'train' if train else 'eval'}", sd_, x, train)
    meta["stft"] = {"kwargs": skw, "n_features": n_feat, "name": sd_.name,
                    "keys": list(sd_.state_dict().keys())}
    rows = []
    for name, mod in ref_d.STFTDiscriminator().named_modules():
        if isinstance(mod, torch.nn.Conv2d):
            rows.append({"name": name, "cin": mod.in_channels, "cout": mod.out_channels, "k": list(mod.kernel_size),
                         "stride": list(mod.stride), "padding": list(mod.padding)})
    meta["stft_default_layers"] = rows

    # ---- the STFT front end alone (torch.stft as the reference calls it) ---------------------------
    torch.manual_seed(9)
    for win, length in ((64, 500), (256, 1024), (1024, 2048)):
        x = torch.randn(2, length)
        y = torch.stft(x, n_fft=win, hop_length=win // 4, win_length=win, normalized=True, return_complex=False,
                       onesided=False)
        g[f"stft_only_{win}/x"] = x.numpy()
        g[f"stft_only_{win}/y"] = y.numpy()          # (B, F, T, 2)

    # ---- the loss ----------------------------------------------------------------------------------
    torch.manual_seed(10)
    blk = ref_d.WaveFormDiscriminator(1, n_blocks=2)
    # shrink: replace the blocks by reduced-width ones (same class, same wiring)
    blk.layers = torch.nn.ModuleList([ref_d.WaveformDiscriminatorBlock(1, scale=s, **wkw) for s in (1, 2)])
    orig = 0.3 * torch.randn(1, 1, 16384)
    rec = orig + 0.05 * torch.randn_like(orig)
    for tag, disc in (("wave", blk), ("stft", ref_d.STFTDiscriminator(**skw))):
        disc.train()
        for k, v in _np(disc.state_dict()).items():
            g[f"loss_{tag}/sd/{k}"] = v
        gl, dl = ref_d.discriminator_generator_loss(orig, rec, disc)
        g[f"loss_{tag}/generator_loss"] = gl.detach().numpy()
        g[f"loss_{tag}/discriminator_loss"] = dl.detach().numpy()
        gl2, dl2 = ref_d.discriminator_generator_loss(orig, rec, disc, feature_multipier=3.0, scale_feature_loss=False)
        g[f"loss_{tag}/generator_loss_unscaled_fm3"] = gl2.detach().numpy()
        g[f"loss_{tag}/discriminator_loss_2nd_call"] = dl2.detach().numpy()
    g["loss/original"] = orig.numpy()
    g["loss/reconstruction"] = rec.numpy()

    np.savez_compressed(os.path.join(OUT, "g7_discriminators.npz"), **g)
    with open(os.path.join(OUT, "meta_g7.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("g7_discriminators.npz", os.path.getsize(os.path.join(OUT, "g7_discriminators.npz")), len(g), "arrays")


if __name__ == "__main__":
    main()
