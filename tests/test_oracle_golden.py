"""The oracle restatement against the reference's own outputs (tests/golden/*)."""
import numpy as np
import pytest
import torch

from oracle import attention as oattn
from oracle import codec, wavelets as owv
from tests.helpers import load_meta, load_npz, max_abs, sub_sd

TOL = 2e-6


def _spec(kwargs):
    return codec.CodecSpec(in_channels=kwargs["in_channels"], n_blocks=kwargs["n_blocks"],
                           first_block_channels=kwargs["first_block_channels"],
                           codebook_dim=kwargs["codebook_dim"], strides=kwargs["strides"],
                           wavelet_decoders=kwargs["wavelet_decoders"],
                           input_format=kwargs["input_format"])


@pytest.mark.parametrize("fixture,key", [("g1_tiny_vqae.npz", "g1"), ("g1b_tiny_wavelet_vqae.npz", "g1b")])
def test_vqae_every_stage(fixture, key):
    blob, meta = load_npz(fixture), load_meta()[key]
    spec = _spec(meta["kwargs"])
    sd = sub_sd(blob, "sd/")
    x = torch.from_numpy(blob["x"])
    xin = x.transpose(1, 2) if spec.input_format == "b l c" else x
    enc = codec.encoder_stages(xin, sd, spec)
    assert len(enc) == spec.n_blocks + 2
    for i, e in enumerate(enc):
        assert e.shape == blob[f"enc_stage_{i}"].shape
        assert max_abs(e, blob[f"enc_stage_{i}"]) < TOL, f"enc stage {i}"
    dec = codec.decoder_stages(torch.from_numpy(blob[f"enc_stage_{len(enc) - 1}"]), sd, spec)
    for i, d in enumerate(dec):
        assert d.shape == blob[f"dec_stage_{i}"].shape
        assert max_abs(d, blob[f"dec_stage_{i}"]) < TOL, f"dec stage {i}"
    # end to end with the quantiser bypassed == the reference with its pass-through placeholder
    y = codec.decode_latents(codec.encode_latents(x, sd, spec), sd, spec)
    assert y.shape == blob["y"].shape
    assert max_abs(y, blob["y"]) < TOL


def test_wavelet_block_position():
    meta = load_meta()["g1b"]
    spec = _spec(meta["kwargs"])
    assert [n for n in range(1, spec.n_blocks + 1) if spec.decoder_is_wavelet(n)] == meta["wavelet_block"]


def test_om_wav_realistic_input():
    blob, g1 = load_npz("g5_om_wav.npz"), load_npz("g1_tiny_vqae.npz")
    spec = _spec(load_meta()["g1"]["kwargs"])
    sd = sub_sd(g1, "sd/")
    x = torch.from_numpy(blob["x"])
    z = codec.encoder_stages(x, sd, spec)[-1]
    assert max_abs(z, blob["enc_stage_5"]) < TOL
    assert max_abs(codec.decoder_stages(z, sd, spec)[-1], blob["y"]) < TOL


def test_primitives():
    blob, cases = load_npz("g2_primitives.npz"), load_meta()["g2"]
    assert len(cases) >= 25
    for c in cases:
        n = c["name"]
        sd = sub_sd(blob, f"{n}/sd/")
        x, want = torch.from_numpy(blob[f"{n}/x"]), blob[f"{n}/y"]
        if c["kind"] == "conv":
            w, b = codec.conv_params(sd, "conv.")
            y = codec.causal_conv1d(x, w, b, stride=c["stride"], dilation=c["dilation"])
        elif c["kind"] == "convt":
            w, b = codec.conv_params(sd, "conv.")
            y = codec.causal_conv_t1d(x, w, b, stride=c["stride"])
        elif c["kind"] == "upconv":
            w, b = codec.conv_params(sd, "conv.")
            y = codec.upsample_conv1d(x, w, b, c["stride"])
        elif c["kind"] == "res":
            y = codec.residual_block(x, sd, "", c["dilation"])
        elif c["kind"] == "encblock":
            y = x
            for j, d in enumerate((1, 3, 9)):
                y = codec.leaky(codec.residual_block(y, sd, f"layers.{j}.0.", d))
            w, b = codec.conv_params(sd, "layers.3.0.conv.")
            y = codec.leaky(codec.causal_conv1d(y, w, b, stride=c["stride"]))
        elif c["kind"] in ("decblock", "decblock_convt"):
            w, b = codec.conv_params(sd, "in_conv.0.conv.")
            if c["kind"] == "decblock":
                y = codec.upsample_conv1d(x, w, b, c["stride"])
            else:
                y = codec.causal_conv_t1d(x, w, b, stride=c["stride"])
            y = codec.leaky(y)
            for j, d in enumerate((1, 3, 9)):
                y = codec.leaky(codec.residual_block(y, sd, f"layers.{j}.0.", d))
        else:
            raise AssertionError(c["kind"])
        assert y.shape == want.shape, n
        assert max_abs(y, want) < TOL, n


def test_causal_pads_known_answers():
    # vae.py:354 / SURVEY 8c: 72000 -> 225 frames with strides (2,4,5,8); ragged lengths round up
    length = 72000
    for s in (2, 4, 5, 8):
        left, right = codec.causal_pads(length, 2 * s + 1, s, 1)
        assert (left, right) == (s + 1, 0)
        length //= s
    assert length == 225
    assert codec.causal_pads(51, 5, 2, 1) == (3, 1)
    assert codec.causal_pads(41, 7, 1, 3) == (18, 0)


def test_attention_block():
    blob, meta = load_npz("g3_attention.npz"), load_meta()["g3"]
    sd = sub_sd(blob, "sd/")
    for name in ("full", "crop"):
        x = torch.from_numpy(blob[f"{name}/x"])
        assert max_abs(oattn.attention(x, sd, "layers.0.0.", meta["heads"]), blob[f"{name}/attn"]) < 5e-6
        assert max_abs(oattn.feed_forward(x, sd, "layers.0.1."), blob[f"{name}/ffn"]) < 5e-6
        assert max_abs(oattn.transformer(x, sd, meta["heads"]), blob[f"{name}/y"]) < 1e-5
    m = blob["alibi_h8_t16"]
    assert m.shape == (1, 8, 16, 16)
    assert max_abs(oattn.alibi_bias(8, 16, 16), m[0]) == 0.0
    assert np.array_equal(oattn.alibi_slopes(8).numpy(), blob["alibi_h8_t225_crop40_slopes"])


def test_multires_and_wavelets():
    blob, cases = load_npz("g4_wavelets.npz"), load_meta()["g4"]
    sd = sub_sd(blob, "multires/sd/")
    y = owv.multires_conv(torch.from_numpy(blob["multires/x"]), sd["h0"], sd["h1"], sd["w"], depth=4)
    assert max_abs(y, blob["multires/y"]) < TOL
    sd = sub_sd(blob, "msblock/sd/")
    y = owv.multires_scale_block(torch.from_numpy(blob["msblock/x"]), sd["multires_conv.h0"],
                                 sd["multires_conv.h1"], sd["multires_conv.w"], 3,
                                 sd["conv.weight"], sd["conv.bias"], 3)
    assert max_abs(y, blob["msblock/y"]) < TOL
    for c in cases:
        n = c["name"]
        y = owv.wavelet_layer(torch.from_numpy(blob[f"{n}/x"]), sub_sd(blob, f"{n}/sd/"), "", c["scale"])
        assert y.shape == blob[f"{n}/y"].shape
        assert max_abs(y, blob[f"{n}/y"]) < TOL, n
    y = owv.wavelet_layer(torch.from_numpy(blob["wavelet_default/x"]), sub_sd(blob, "wavelet_default/sd/"), "", 2)
    assert max_abs(y, blob["wavelet_default/y"]) < TOL


def test_shape_table_known_answers():
    g6 = load_meta()["g6"]
    spec = codec.CodecSpec(n_blocks=4, strides=(2, 4, 5, 8), wavelet_decoders=False, input_format="n c l")
    sd = codec.init_state_dict(spec)
    assert list(sd.keys()) != [] and set(sd.keys()) == set(g6["state_dict_keys"])
    assert sum(v.numel() for k, v in sd.items() if k.startswith("encoders")) == g6["params_encoders"]
    assert sum(v.numel() for k, v in sd.items() if k.startswith("decoders")) == g6["params_decoders"]
    assert g6["output"] == [1, 1, 72000]
    enc_macs = sum(r["macs"] for r in g6["layers"] if r["name"].startswith("encoders"))
    dec_macs = sum(r["macs"] for r in g6["layers"] if r["name"].startswith("decoders"))
    assert abs(enc_macs / 72000 - 195194) < 1 and abs(dec_macs / 72000 - 316026) < 1


def _depthwise_cases():
    blob = load_npz("g8_depthwise.npz")
    for name, dil in (("res_d3", 3), ("res_d9", 9)):
        yield name, blob, lambda x, sd, d=dil: codec.residual_block(x, sd, "", d)

    def enc(x, sd):
        y = x
        for j, d in enumerate((1, 3, 9)):
            y = codec.leaky(codec.residual_block(y, sd, f"layers.{j}.0.", d))
        w, b = codec.conv_params(sd, "layers.3.0.conv.")
        return codec.leaky(codec.causal_conv1d(y, w, b, stride=4))

    def dec(x, sd):
        w, b = codec.conv_params(sd, "in_conv.0.conv.")
        y = codec.leaky(codec.upsample_conv1d(x, w, b, 5))
        for j, d in enumerate((1, 3, 9)):
            y = codec.leaky(codec.residual_block(y, sd, f"layers.{j}.0.", d))
        return y

    yield "encblock", blob, enc
    yield "decblock", blob, dec


def test_depthwise_variant_matches_the_reference():
    """depthwise=True residual blocks (vae.py:103-105), alone and inside encoder / decoder blocks (golden G8)."""
    n = 0
    for name, blob, fn in _depthwise_cases():
        sd = sub_sd(blob, f"{name}/sd/")
        assert any(k.startswith("conv1.0.") or ".conv1.0." in k for k in sd)
        y = fn(torch.from_numpy(blob[f"{name}/x"]), sd)
        assert max_abs(y, blob[f"{name}/y"]) < TOL, name
        n += 1
    assert n == 4
