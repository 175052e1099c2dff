"""Generate the golden fixtures of ``tests/golden/`` from the REFERENCE.

Run once, in the build container only (``/root/reference`` does not exist on
the GPU box):  ``python tests/golden/make_goldens.py``.

The reference's modules are imported from ``/root/reference/networks`` as they
are.  Two of their top-level imports are not installed here -- ``torchaudio``
(unused by the conv / attention / wavelet classes) and ``som_quantizer`` (the
external RVQ, absent from the reference tree) -- so empty placeholder modules
are registered for those two names; the quantiser placeholder is a pass-through
so that encoder and decoder can be exercised around it.  Nothing from the
reference is written into this repository except numeric inputs / outputs.

This script imports nothing from ``oracle/`` or the product package: fixtures
are the reference's word alone.
"""
from __future__ import annotations

import json
import os
import struct
import sys
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import numpy as np
import torch

REF = "/root/reference/networks"
OUT = os.path.dirname(os.path.abspath(__file__))


def _install_placeholders():
    sys.modules.setdefault("torchaudio", types.ModuleType("torchaudio"))
    som = types.ModuleType("som_quantizer")

    class ResidualQuantizer(torch.nn.Module):  # pass-through stand-in for the absent RVQ
        def __init__(self, **kw):
            super().__init__()
            self.num_quantizers = kw.get("num_quantizers", 1)

        def forward(self, x, codebook_n=None, update_codebook=False, prioritize_early=False):
            return x, None, torch.zeros(())

    def tuple_checker(item, length):
        if isinstance(item, (int, float, str)):
            return [item] * length
        assert len(item) == length
        return item

    som.ResidualQuantizer = ResidualQuantizer
    som.tuple_checker = tuple_checker
    sys.modules.setdefault("som_quantizer", som)
    # matplotlib is imported by utils/wavelets at module level; use a non-GUI backend
    os.environ.setdefault("MPLBACKEND", "Agg")


def _np(sd):
    return {k: v.detach().cpu().numpy() for k, v in sd.items()}


def _stage_outputs(model, x):
    """Outputs of every entry of model.encoders / model.decoders (quantiser bypassed)."""
    outs = {}
    with torch.no_grad():
        h = model.rearrange_in(x)
        for i, enc in enumerate(model.encoders):
            h = enc(h)
            outs[f"enc_stage_{i}"] = h.numpy().copy()
        for i, dec in enumerate(model.decoders):
            h = dec(h)
            outs[f"dec_stage_{i}"] = h.numpy().copy()
        outs["y"] = model.rearrange_out(h).numpy().copy()
    return outs


def read_wav_float32(path):
    """Minimal RIFF/WAVE float32 reader (om.wav is IEEE-float, 2 ch, 16 kHz)."""
    with open(path, "rb") as f:
        data = f.read()
    assert data[:4] == b"RIFF" and data[8:12] == b"WAVE"
    pos, fmt, pcm = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            fmt = struct.unpack("<HHIIHH", body[:16])
        elif cid == b"data":
            pcm = body
        pos += 8 + size + (size & 1)
    tag, ch, rate, _, _, bits = fmt
    assert tag in (3, 0xFFFE) and bits == 32, fmt
    arr = np.frombuffer(pcm, dtype="<f4").reshape(-1, ch).T.copy()
    return arr, rate


def main():
    _install_placeholders()
    sys.path.insert(0, REF)
    import vae          # noqa: E402  (reference)
    import transformers as ref_tf  # noqa: E402  (reference's networks/transformers.py)
    import wavelets     # noqa: E402  (reference)

    torch.set_num_threads(4)
    meta = {"torch": torch.__version__, "generator": "tests/golden/make_goldens.py"}

    # ---- G1: tiny VQAE, every stage --------------------------------------------------
    torch.manual_seed(0)
    kw = dict(in_channels=1, n_blocks=4, strides=(2, 4, 5, 8), first_block_channels=4,
              codebook_dim=16, num_quantizers=1, codebook_size=16, input_format="n c l",
              wavelet_decoders=False)
    model = vae.CausalVQAE(**kw).eval()
    x = 0.1 * torch.randn(2, 1, 1600)
    g1 = {f"sd/{k}": v for k, v in _np(model.state_dict()).items()}
    g1["x"] = x.numpy()
    g1.update(_stage_outputs(model, x))
    np.savez_compressed(os.path.join(OUT, "g1_tiny_vqae.npz"), **g1)
    meta["g1"] = {"kwargs": {k: (list(v) if isinstance(v, tuple) else v) for k, v in kw.items()},
                  "seed": 0}

    # ---- G5: same tiny model on 1 s of om.wav (mono mix, vae.py:378) ------------------
    wav, rate = read_wav_float32(os.path.join(REF, "om.wav"))
    mono = torch.from_numpy(wav).mean(dim=0, keepdim=True).unsqueeze(0)[:, :, :16000]
    g5 = {"x": mono.numpy()}
    g5.update({k: v for k, v in _stage_outputs(model, mono).items()
               if k in ("enc_stage_5", "y")})
    np.savez_compressed(os.path.join(OUT, "g5_om_wav.npz"), **g5)
    meta["g5"] = {"rate": rate, "channels": int(wav.shape[0]), "frames": int(wav.shape[1]),
                  "model": "g1"}

    # ---- G1b: tiny VQAE with a wavelet decoder block + stereo + 'b l c' ---------------
    torch.manual_seed(1)
    kwb = dict(in_channels=2, n_blocks=4, strides=(2, 4, 5, 8), first_block_channels=4,
               codebook_dim=16, num_quantizers=1, codebook_size=16, input_format="b l c",
               wavelet_decoders=[False, True, False, False])
    modelb = vae.CausalVQAE(**kwb).eval()
    xb = 0.1 * torch.randn(1, 960, 2)
    g1b = {f"sd/{k}": v for k, v in _np(modelb.state_dict()).items()}
    g1b["x"] = xb.numpy()
    g1b.update(_stage_outputs(modelb, xb))
    np.savez_compressed(os.path.join(OUT, "g1b_tiny_wavelet_vqae.npz"), **g1b)
    meta["g1b"] = {"kwargs": {k: (list(v) if isinstance(v, tuple) else v) for k, v in kwb.items()},
                   "seed": 1,
                   "wavelet_block": [i for i, d in enumerate(modelb.decoders)
                                     if getattr(d, "wavelet", False)]}

    # ---- G2: primitives at awkward sizes ----------------------------------------------
    g2, cases = {}, []
    torch.manual_seed(2)

    def add_case(name, mod, xin, **info):
        mod.eval()
        with torch.no_grad():
            y = mod(xin)
        for k, v in _np(mod.state_dict()).items():
            g2[f"{name}/sd/{k}"] = v
        g2[f"{name}/x"] = xin.numpy()
        g2[f"{name}/y"] = y.numpy()
        cases.append(dict(name=name, **info))

    i = 0
    for (cin, cout, k, s, d, length) in [(3, 5, 7, 1, 1, 50), (3, 5, 7, 1, 3, 41), (4, 4, 7, 1, 9, 64),
                                         (4, 6, 5, 2, 1, 51), (4, 6, 9, 4, 1, 50), (2, 3, 11, 5, 1, 53),
                                         (2, 3, 17, 8, 1, 61), (5, 4, 3, 1, 1, 17), (4, 4, 1, 1, 1, 23),
                                         (1, 4, 7, 1, 1, 30), (4, 1, 7, 1, 1, 30), (3, 4, 5, 3, 2, 40)]:
        add_case(f"conv{i}", vae.CausalConv1d(cin, cout, k, dilation=d, stride=s),
                 torch.randn(2, cin, length), kind="conv", cin=cin, cout=cout, k=k, stride=s,
                 dilation=d)
        i += 1
    for j, (cin, cout, k, s, length) in enumerate([(4, 6, 7, 1, 20), (3, 2, 5, 2, 11), (2, 3, 9, 4, 7),
                                                   (3, 3, 17, 8, 5)]):
        add_case(f"convt{j}", vae.CausalConvT1d(cin, cout, k, stride=s), torch.randn(2, cin, length),
                 kind="convt", cin=cin, cout=cout, k=k, stride=s)
    for j, (cin, cout, s, length) in enumerate([(4, 3, 2, 9), (4, 2, 4, 7), (3, 3, 5, 6), (2, 3, 8, 5)]):
        add_case(f"upconv{j}", vae.CausalUpsampleConv1d(cin, cout, 2 * s + 1, stride=s),
                 torch.randn(2, cin, length), kind="upconv", cin=cin, cout=cout, k=2 * s + 1, stride=s)
    for j, (c, d, length) in enumerate([(4, 1, 33), (4, 3, 40), (6, 9, 70)]):
        add_case(f"res{j}", vae.CausalResidualBlock1d(c, c, dilation=d), torch.randn(2, c, length),
                 kind="res", c=c, dilation=d)
    add_case("encblock0", vae.CausalEncoderBlock(4, 8, 4), torch.randn(2, 4, 84), kind="encblock",
             cin=4, cout=8, stride=4)
    add_case("decblock0", vae.CausalDecoderBlock(8, 4, 5), torch.randn(2, 8, 13), kind="decblock",
             cin=8, cout=4, stride=5)
    add_case("decblock_convt", vae.CausalDecoderBlock(8, 4, 4, upsample=False), torch.randn(2, 8, 11),
             kind="decblock_convt", cin=8, cout=4, stride=4)
    np.savez_compressed(os.path.join(OUT, "g2_primitives.npz"), **g2)
    meta["g2"] = cases

    # ---- G3: attention ------------------------------------------------------------------
    torch.manual_seed(3)
    tf = ref_tf.Transformer(64, depth=1, heads=4, head_dim=16, context_x=50).eval()
    # LayerNorm affine away from identity so it is exercised
    with torch.no_grad():
        for p in tf.parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    g3 = {f"sd/{k}": v for k, v in _np(tf.state_dict()).items()}
    for name, t in (("full", 50), ("crop", 37)):
        xin = torch.randn(2, t, 64)
        with torch.no_grad():
            g3[f"{name}/y"] = tf(xin).numpy()
            g3[f"{name}/attn"] = tf.layers[0][0](xin).numpy()
            g3[f"{name}/ffn"] = tf.layers[0][1](xin).numpy()
        g3[f"{name}/x"] = xin.numpy()
    g3["alibi_h8_t16"] = ref_tf.Alibi(16, n_heads=8).get_M().numpy()
    g3["alibi_h8_t225_crop40_slopes"] = ref_tf.Alibi(225, n_heads=8).head_scalars.numpy()
    np.savez_compressed(os.path.join(OUT, "g3_attention.npz"), **g3)
    meta["g3"] = {"dim": 64, "heads": 4, "head_dim": 16, "context_x": 50}

    # ---- G4: multires + wavelet layers ----------------------------------------------------
    torch.manual_seed(4)
    g4, wcases = {}, []
    mr = wavelets.CausalMultiresConv1d(8, 3, 4).eval()
    xin = torch.randn(2, 8, 70)
    with torch.no_grad():
        g4["multires/y"] = mr(xin).numpy()
    g4["multires/x"] = xin.numpy()
    for k, v in _np(mr.state_dict()).items():
        g4[f"multires/sd/{k}"] = v
    msb = wavelets.MultiresScaleBlock(6, 4, scale_factor=3, kernel_size=3, multires_depth=3).eval()
    xin = torch.randn(2, 6, 25)
    with torch.no_grad():
        g4["msblock/y"] = msb(xin).numpy()
    g4["msblock/x"] = xin.numpy()
    for k, v in _np(msb.state_dict()).items():
        g4[f"msblock/sd/{k}"] = v
    for s in (2, 4, 5, 8):
        cin, cout = 6, 3
        # wired exactly as vae.py:167-173 does
        wl = wavelets.WaveletLayer(cin, cout * 4, out_channels=cout, scale_factor=s,
                                   wavelet_kernel_size=2 * s + 1, n_points=2 * s * 4,
                                   channelwise_scale=True).eval()
        with torch.no_grad():
            wl.wavelet_scale.mul_(1 + 0.2 * torch.rand_like(wl.wavelet_scale))
        xin = torch.randn(2, cin, 11)
        with torch.no_grad():
            g4[f"wavelet_s{s}/y"] = wl(xin).numpy()
        g4[f"wavelet_s{s}/x"] = xin.numpy()
        for k, v in _np(wl.state_dict()).items():
            g4[f"wavelet_s{s}/sd/{k}"] = v
        wcases.append({"name": f"wavelet_s{s}", "scale": s, "cin": cin, "cout": cout})
    # a non-channelwise, default-argument instance (exercises other n_points/fold ratios)
    wl = wavelets.WaveletLayer(4, 8, scale_factor=2, channelwise_scale=False).eval()
    xin = torch.randn(1, 4, 9)
    with torch.no_grad():
        g4["wavelet_default/y"] = wl(xin).numpy()
    g4["wavelet_default/x"] = xin.numpy()
    for k, v in _np(wl.state_dict()).items():
        g4[f"wavelet_default/sd/{k}"] = v
    np.savez_compressed(os.path.join(OUT, "g4_wavelets.npz"), **g4)
    meta["g4"] = wcases

    # ---- G6: shape / MAC table of the default ("config S") model ---------------------------
    torch.manual_seed(0)
    big = vae.CausalVQAE(in_channels=1, n_blocks=4, strides=(2, 4, 5, 8), num_quantizers=8,
                         codebook_size=1024, codebook_dim=512, input_format="n c l",
                         wavelet_decoders=False).eval()
    rows = []

    def hook(name):
        def fn(mod, inp, out):
            w = mod.weight
            if isinstance(mod, torch.nn.ConvTranspose1d):
                cin, cout, k = w.shape
                macs = cin * cout * k * inp[0].shape[-1]
            else:
                cout, cin, k = w.shape
                macs = cin * cout * k * out.shape[-1]
            rows.append({"name": name, "type": type(mod).__name__, "cin": int(cin), "cout": int(cout),
                         "k": int(k), "stride": int(mod.stride[0]), "dilation": int(mod.dilation[0]),
                         "lin": int(inp[0].shape[-1]), "lout": int(out.shape[-1]), "macs": int(macs)})
        return fn

    for name, mod in big.named_modules():
        if isinstance(mod, (torch.nn.Conv1d, torch.nn.ConvTranspose1d)):
            mod.register_forward_hook(hook(name))
    with torch.no_grad():
        yb, _, _ = big(torch.zeros(1, 1, 72000))
    n_enc = sum(p.numel() for p in big.encoders.parameters())
    n_dec = sum(p.numel() for p in big.decoders.parameters())
    meta["g6"] = {"input": [1, 1, 72000], "output": list(yb.shape), "layers": rows,
                  "params_encoders": n_enc, "params_decoders": n_dec,
                  "state_dict_keys": list(big.state_dict().keys())}

    with open(os.path.join(OUT, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1)
    for fn in sorted(os.listdir(OUT)):
        print(fn, os.path.getsize(os.path.join(OUT, fn)))


if __name__ == "__main__":
    main()
