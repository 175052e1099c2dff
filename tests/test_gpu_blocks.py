"""GPU parity of the attention bottleneck (config 3) and the wavelet / multires
layers (config 4) against the reference's golden vectors and the CPU oracle."""
import pytest
import torch

from audio_generation_amd import ops
from audio_generation_amd.transformers import Alibi, Transformer, TransformerBottleneck
from audio_generation_amd.vae import CausalVQAE
from audio_generation_amd.wavelets import CausalMultiresConv1d, MultiresScaleBlock, WaveletLayer
from oracle import attention as oattn
from oracle import codec
from oracle import wavelets as owv
from tests.helpers import load_meta, load_npz, max_abs, rms, sub_sd

pytestmark = pytest.mark.gpu
DEV = "cuda"


# ------------------------------------------------------------------------- attention
def test_golden_transformer_block():
    blob, meta = load_npz("g3_attention.npz"), load_meta()["g3"]
    tf = Transformer(meta["dim"], depth=1, heads=meta["heads"], head_dim=meta["head_dim"],
                     context_x=meta["context_x"])
    tf.load_state_dict(sub_sd(blob, "sd/"))
    tf = tf.to(DEV).eval()
    for name in ("full", "crop"):
        x = torch.from_numpy(blob[f"{name}/x"]).to(DEV)
        with torch.no_grad():
            attn = tf.layers[0][0](x)
            ffn = tf.layers[0][1](x)
            y = tf(x)
        assert max_abs(attn.cpu(), blob[f"{name}/attn"]) < 1e-5, name
        assert max_abs(ffn.cpu(), blob[f"{name}/ffn"]) < 1e-5, name
        assert max_abs(y.cpu(), blob[f"{name}/y"]) < 2e-5, name
    m = Alibi(16, n_heads=8).get_M()
    assert torch.equal(m, torch.from_numpy(blob["alibi_h8_t16"]))
    assert torch.equal(Alibi(225, n_heads=8).head_scalars, torch.from_numpy(blob["alibi_h8_t225_crop40_slopes"]))


@pytest.mark.parametrize("t,b", [(225, 4), (75, 2), (1, 1), (33, 3), (256, 1)])
def test_transformer_config3_shape(t, b):
    """Transformer(512, depth=1, heads=8, head_dim=64) as in SURVEY 8d (C3), fp32."""
    sd = oattn.init_state_dict(512, 8, 64, seed=t)
    tf = Transformer(512, depth=1, heads=8, head_dim=64, context_x=256)
    tf.load_state_dict(sd)
    tf = tf.to(DEV).eval()
    x = torch.randn(b, t, 512, generator=torch.Generator().manual_seed(t))
    want = oattn.transformer(x, sd, 8)
    with torch.no_grad():
        y = tf(x.to(DEV))
        y_bct = tf.run_bct(x.to(DEV).transpose(1, 2).contiguous())
    assert torch.equal(y, y_bct.transpose(1, 2))
    err, scale = max_abs(y.cpu(), want), float(want.abs().max())
    assert err < 2e-5 * max(1.0, scale), (err, scale)
    assert rms(y.cpu(), want) < 2e-6 * max(1.0, scale)


def test_attention_odd_head_dim_and_depth2():
    sd = oattn.init_state_dict(60, 3, 20, depth=2, seed=9)   # head_dim 20, inner 60, two layers
    tf = Transformer(60, depth=2, heads=3, head_dim=20, context_x=64)
    tf.load_state_dict(sd)
    tf = tf.to(DEV).eval()
    x = torch.randn(2, 50, 60, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        y = tf(x.to(DEV))
    assert max_abs(y.cpu(), oattn.transformer(x, sd, 3, depth=2)) < 3e-5


def test_context_overflow_raises_like_the_reference():
    tf = Transformer(64, depth=1, heads=4, head_dim=16, context_x=32).to(DEV).eval()
    with pytest.raises(Exception):
        tf(torch.randn(1, 40, 64, device=DEV))


def test_bottleneck_swap_in_vqae():
    """replace_quantizer(TransformerBottleneck) -- the config-3 wiring (vae.py:347-348)."""
    torch.manual_seed(5)
    kw = dict(in_channels=1, n_blocks=4, strides=(2, 4, 5, 8), num_quantizers=2, codebook_size=32,
              codebook_dim=512, input_format="n c l", wavelet_decoders=False)
    model = CausalVQAE(**kw).eval()
    tf = Transformer(512, depth=1, heads=8, head_dim=64, context_x=225).eval()
    model.replace_quantizer(TransformerBottleneck(tf))
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    tsd = {k[len("quantizer.transformer."):]: v for k, v in sd.items() if k.startswith("quantizer.transformer.")}
    spec = codec.CodecSpec(in_channels=1, n_blocks=4, strides=(2, 4, 5, 8), codebook_dim=512,
                           wavelet_decoders=False, input_format="n c l")
    x = 0.1 * torch.randn(2, 1, 6400)
    z = codec.encode_latents(x, sd, spec)
    want = codec.decode_latents(oattn.transformer(z, tsd, 8), sd, spec)
    model = model.to(DEV)
    with torch.no_grad():
        y, loss, index = model(x.to(DEV))
    assert index is None and float(loss) == 0.0
    assert rms(y.cpu(), want) < 1e-4 and max_abs(y.cpu(), want) < 1e-4


# -------------------------------------------------------------------- wavelets / multires
def test_golden_multires_and_wavelet_layers():
    blob, cases = load_npz("g4_wavelets.npz"), load_meta()["g4"]
    mr = CausalMultiresConv1d(8, 3, 4)
    mr.load_state_dict(sub_sd(blob, "multires/sd/"))
    with torch.no_grad():
        y = mr.to(DEV)(torch.from_numpy(blob["multires/x"]).to(DEV))
    assert max_abs(y.cpu(), blob["multires/y"]) < 5e-6
    msb = MultiresScaleBlock(6, 4, scale_factor=3, kernel_size=3, multires_depth=3)
    msb.load_state_dict(sub_sd(blob, "msblock/sd/"))
    with torch.no_grad():
        y = msb.to(DEV)(torch.from_numpy(blob["msblock/x"]).to(DEV))
    assert y.shape == blob["msblock/y"].shape and max_abs(y.cpu(), blob["msblock/y"]) < 5e-6
    for c in cases:
        n, s = c["name"], c["scale"]
        wl = WaveletLayer(c["cin"], c["cout"] * 4, out_channels=c["cout"], scale_factor=s,
                          wavelet_kernel_size=2 * s + 1, n_points=2 * s * 4, channelwise_scale=True)
        wl.load_state_dict(sub_sd(blob, f"{n}/sd/"))
        with torch.no_grad():
            y = wl.to(DEV)(torch.from_numpy(blob[f"{n}/x"]).to(DEV))
        assert y.shape == blob[f"{n}/y"].shape, n
        assert max_abs(y.cpu(), blob[f"{n}/y"]) < 1e-5, n
    wl = WaveletLayer(4, 8, scale_factor=2, channelwise_scale=False)
    wl.load_state_dict(sub_sd(blob, "wavelet_default/sd/"))
    with torch.no_grad():
        y = wl.to(DEV)(torch.from_numpy(blob["wavelet_default/x"]).to(DEV))
    assert max_abs(y.cpu(), blob["wavelet_default/y"]) < 1e-5


def test_golden_tiny_wavelet_vqae_stage_by_stage():
    """G1b: stereo, 'b l c' input, wavelet decoder in the stride-5 block."""
    blob, meta = load_npz("g1b_tiny_wavelet_vqae.npz"), load_meta()["g1b"]["kwargs"]
    model = CausalVQAE(**{**meta, "strides": tuple(meta["strides"])})
    model.load_state_dict(sub_sd(blob, "sd/"), strict=False)
    model = model.to(DEV).eval()
    assert [i for i, d in enumerate(model.decoders) if getattr(d, "wavelet", False)] == load_meta()["g1b"]["wavelet_block"]
    x = torch.from_numpy(blob["x"]).to(DEV)
    with torch.no_grad():
        h = model.encoders[0][1](model.rearrange_in(x))
        assert max_abs(h.cpu(), blob["enc_stage_0"]) < 5e-6
        for i, enc in enumerate(list(model.encoders)[1:], start=1):
            h = enc(h)
            assert max_abs(h.cpu(), blob[f"enc_stage_{i}"]) < 5e-6, f"enc stage {i}"
        for i, dec in enumerate(model.decoders):
            h = dec(h)
            assert max_abs(h.cpu(), blob[f"dec_stage_{i}"]) < 1e-5, f"dec stage {i}"
        y = model.rearrange_out(h)
    assert max_abs(y.cpu(), blob["y"]) < 1e-5


@pytest.mark.parametrize("c,k,depth,length", [(16, 3, 6, 5000), (5, 13, 4, 1500), (3, 2, 1, 77)])
def test_multires_larger_shapes(c, k, depth, length):
    gen = torch.Generator().manual_seed(depth)
    mr = CausalMultiresConv1d(c, k, depth)
    x = torch.randn(2, c, length, generator=gen)
    want = owv.multires_conv(x, mr.h0.detach(), mr.h1.detach(), mr.w.detach(), depth)
    with torch.no_grad():
        y = mr.to(DEV)(x.to(DEV))
    assert max_abs(y.cpu(), want) < 1e-5


def test_wavelet_vqae_config4_topology():
    """48 kHz-style stereo model with the reference-wired wavelet block at full channel widths."""
    torch.manual_seed(8)
    kw = dict(in_channels=2, n_blocks=4, strides=(2, 4, 5, 8), num_quantizers=2, codebook_size=64,
              codebook_dim=512, input_format="n c l", wavelet_decoders=[False, True, False, False])
    model = CausalVQAE(**kw).eval()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    spec = codec.CodecSpec(in_channels=2, n_blocks=4, strides=(2, 4, 5, 8), codebook_dim=512,
                           wavelet_decoders=[False, True, False, False], input_format="n c l")
    x = 0.1 * torch.randn(1, 2, 6400)
    z = codec.encode_latents(x, sd, spec)
    want = codec.decode_latents(z, sd, spec)           # quantiser bypassed: conv + wavelet stacks only
    model = model.to(DEV)
    with torch.no_grad():
        zg = model._run_encoders(x.to(DEV))
        y = model.decode(zg)
    assert rms(zg.cpu().transpose(1, 2), z) < 1e-5
    assert rms(y.cpu(), want) < 1e-4


def test_config4_multires_in_encoder_and_decoder_at_model_level():
    """BASELINE config 4 as written: multiresolution layers in encoder AND decoder next to the reference-wired wavelet block,
    48 kHz-style stereo.  The placement is build-defined (CausalVQAE docstring; oracle/codec.py CodecSpec restates it): forward
    against the oracle stage by stage, then one native training backward against the oracle's autograd."""
    torch.manual_seed(18)
    kw = dict(in_channels=2, n_blocks=4, strides=(2, 4, 5, 8), num_quantizers=2, codebook_size=64, codebook_dim=512,
              input_format="n c l", wavelet_decoders=[False, True, False, False],
              multires_encoders=[True, False, True, False], multires_decoders=[False, False, True, True],
              multires_kernel_size=3, multires_depth=4)
    model = CausalVQAE(**kw).eval()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    assert "encoders.1.multires.h0" in sd and "decoders.3.multires.w" in sd and "encoders.2.multires.h0" not in sd
    spec = codec.CodecSpec(in_channels=2, n_blocks=4, strides=(2, 4, 5, 8), codebook_dim=512, input_format="n c l",
                           wavelet_decoders=[False, True, False, False], multires_encoders=[True, False, True, False],
                           multires_decoders=[False, False, True, True], multires_kernel_size=3, multires_depth=4)
    x = 0.1 * torch.randn(1, 2, 6400)
    stages = codec.encoder_stages(x, sd, spec)
    z = stages[-1].transpose(1, 2).contiguous()
    want = codec.decode_latents(z, sd, spec)
    model = model.to(DEV)
    with torch.no_grad():
        h = x.to(DEV)
        for i, enc in enumerate(model.encoders):
            h = enc(h) if i else enc[1].run(h)
            assert rms(h.cpu(), stages[i]) < 1e-5 * max(1.0, float(stages[i].abs().max())), i
        y = model.decode(h)
    assert rms(y.cpu(), want) < 1e-4
    # training: gradients of sum(y * g) w.r.t. a multires filter, a conv weight next to it, and the input
    small = dict(kw, first_block_channels=4, codebook_dim=16)
    m2 = CausalVQAE(**small).train().to(DEV)
    sd2 = {k: v.detach().cpu().clone() for k, v in m2.state_dict().items()}
    spec2 = codec.CodecSpec(in_channels=2, n_blocks=4, strides=(2, 4, 5, 8), codebook_dim=16, first_block_channels=4,
                            input_format="n c l", wavelet_decoders=[False, True, False, False],
                            multires_encoders=[True, False, True, False], multires_decoders=[False, False, True, True],
                            multires_kernel_size=3, multires_depth=4)
    x2 = 0.1 * torch.randn(2, 2, 3200)
    names = ["encoders.1.multires.h1", "encoders.3.multires.w", "decoders.3.multires.h0", "decoders.4.multires.w",
             "encoders.1.layers.3.0.conv.weight_v", "decoders.3.in_conv.0.conv.weight_v"]
    leaves = {k: sd2[k].clone().requires_grad_(True) for k in names}
    sdl = dict(sd2, **leaves)
    xl = x2.clone().requires_grad_(True)
    zl = codec.encoder_stages(xl, sdl, spec2)[-1]
    yl = codec.decoder_stages(zl, sdl, spec2)[-1]
    gout = torch.randn_like(yl)
    (yl * gout).sum().backward()
    xg = x2.to(DEV).requires_grad_(True)
    yg = m2._run_decoders(m2._run_encoders(xg))
    assert yg.requires_grad and rms(yg.detach().cpu(), yl.detach()) < 1e-5
    (yg * gout.to(DEV)).sum().backward()
    params = dict(m2.named_parameters())
    assert max_abs(xg.grad.cpu(), xl.grad) < 2e-3 * float(xl.grad.abs().max())
    for k in names:
        ref = leaves[k].grad
        assert params[k].grad is not None and max_abs(params[k].grad.cpu(), ref) < 2e-3 * max(1e-6, float(ref.abs().max())), k


def test_depthwise_residual_variant_matches_the_reference_goldens():
    """CausalResidualBlock1d / encoder / decoder blocks with depthwise=True (vae.py:103-105; golden G8): the
    per-channel k = 1 conv runs as a grouped AGX_CONV_PADDED layer in front of the dilated conv."""
    from audio_generation_amd.vae import CausalDecoderBlock, CausalEncoderBlock, CausalResidualBlock1d
    from tests.helpers import load_npz, sub_sd
    blob = load_npz("g8_depthwise.npz")
    mods = {"res_d3": CausalResidualBlock1d(6, 6, dilation=3, depthwise=True),
            "res_d9": CausalResidualBlock1d(16, 16, dilation=9, depthwise=True),
            "encblock": CausalEncoderBlock(4, 8, 4, depthwise=True),
            "decblock": CausalDecoderBlock(8, 4, 5, depthwise=True)}
    for name, m in mods.items():
        m.load_state_dict(sub_sd(blob, f"{name}/sd/"), strict=True)
        m = m.to(DEV).eval()
        with torch.no_grad():
            y = m(torch.from_numpy(blob[f"{name}/x"]).to(DEV))
        assert max_abs(y.cpu(), blob[f"{name}/y"]) < 2e-5, name


def test_multires_layers_backward_on_the_hip_kernels():
    """``CausalMultiresConv1d`` / ``MultiresScaleBlock`` (wavelets.py:79-121; never instantiated by the reference,
    vae.py:7): ``agx_multires_backward`` (+ ``agx_group_sum`` and the k = 1 conv's backward kernels for the scale block)
    against the oracle's autograd -- input and every parameter, G4-sized and larger shapes incl. several tiles."""
    torch.manual_seed(9)
    cases = [(CausalMultiresConv1d(8, 3, 4), 8, 90, lambda t, p: owv.multires_conv(t, p["h0"], p["h1"], p["w"], 4)),
             (CausalMultiresConv1d(5, 4, 3), 5, 1500, lambda t, p: owv.multires_conv(t, p["h0"], p["h1"], p["w"], 3)),
             (CausalMultiresConv1d(3, 2, 6), 3, 700, lambda t, p: owv.multires_conv(t, p["h0"], p["h1"], p["w"], 6)),
             (CausalMultiresConv1d(4, 3, 0), 4, 64, lambda t, p: owv.multires_conv(t, p["h0"], p["h1"], p["w"], 0)),
             (MultiresScaleBlock(8, 6, scale_factor=2, kernel_size=3, multires_depth=3), 8, 90,
              lambda t, p: owv.multires_scale_block(t, p["multires_conv.h0"], p["multires_conv.h1"], p["multires_conv.w"], 3,
                                                    p["conv.weight"], p["conv.bias"], 2)),
             (MultiresScaleBlock(16, 32, scale_factor=5, kernel_size=3, multires_depth=4), 16, 1100,
              lambda t, p: owv.multires_scale_block(t, p["multires_conv.h0"], p["multires_conv.h1"], p["multires_conv.w"], 4,
                                                    p["conv.weight"], p["conv.bias"], 5))]
    for mod, c, length, fn in cases:
        mod = mod.to(DEV)
        x = torch.randn(2, c, length)
        leaves = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in mod.named_parameters()}
        xl = x.clone().requires_grad_(True)
        want = fn(xl, leaves)
        gout = torch.randn_like(want)
        (want * gout).sum().backward()
        xg = x.to(DEV).requires_grad_(True)
        got = mod(xg)
        assert got.requires_grad and max_abs(got.detach().cpu(), want.detach()) < 2e-5
        (got * gout.to(DEV)).sum().backward()
        assert max_abs(xg.grad.cpu(), xl.grad) < 2e-5 * max(1.0, float(xl.grad.abs().max()))
        for k, p in mod.named_parameters():
            ref = leaves[k].grad if leaves[k].grad is not None else torch.zeros_like(leaves[k])   # depth 0: unused filters
            scale = max(1.0, float(ref.abs().max()))
            assert p.grad is not None and max_abs(p.grad.cpu(), ref) < 1e-4 * scale, (type(mod).__name__, k)
