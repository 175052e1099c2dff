"""Generate ``g8_depthwise.npz`` from the REFERENCE: the ``depthwise=True`` residual / encoder / decoder blocks
of networks/vae.py (:103-105, 119-202) -- unused by the shipped configs but part of the constructor surface.
Same rules as ``make_goldens.py`` (build container only; numeric inputs / outputs only)."""
from __future__ import annotations

import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_goldens import OUT, REF, _install_placeholders  # noqa: E402


def main():
    _install_placeholders()
    sys.path.insert(0, REF)
    import vae  # noqa: E402  (reference)

    torch.manual_seed(11)
    g = {}

    def add(name, mod, x):
        mod.eval()
        with torch.no_grad():
            y = mod(x)
        for k, v in mod.state_dict().items():
            g[f"{name}/sd/{k}"] = v.detach().numpy().copy()
        g[f"{name}/x"], g[f"{name}/y"] = x.numpy().copy(), y.numpy().copy()

    add("res_d3", vae.CausalResidualBlock1d(6, 6, dilation=3, depthwise=True), torch.randn(2, 6, 41))
    add("res_d9", vae.CausalResidualBlock1d(16, 16, dilation=9, depthwise=True), torch.randn(2, 16, 70))
    add("encblock", vae.CausalEncoderBlock(4, 8, 4, depthwise=True), torch.randn(2, 4, 84))
    add("decblock", vae.CausalDecoderBlock(8, 4, 5, depthwise=True), torch.randn(2, 8, 13))
    np.savez_compressed(os.path.join(OUT, "g8_depthwise.npz"), **g)
    print("g8_depthwise.npz", os.path.getsize(os.path.join(OUT, "g8_depthwise.npz")), len(g), "arrays")


if __name__ == "__main__":
    main()
