"""Parity of the HIP path (through the C ABI) against the CPU oracle and the
reference's golden vectors.  Needs the MI355X: ``pytest -m gpu``.

Tolerances: indices and every integer output bit-exact; fp32 conv outputs
within a few ulp-scale absolute error per layer (stated per test) and the
reconstructed waveform within the 1e-4 RMS budget of BASELINE.json.
"""
import numpy as np
import pytest
import torch

from audio_generation_amd import _lib, ops
from audio_generation_amd.vae import (CausalConv1d, CausalConvT1d, CausalDecoderBlock, CausalEncoderBlock,
                                      CausalResidualBlock1d, CausalUpsampleConv1d, CausalVQAE)
from oracle import codec, neartie, rvq
from tests.helpers import load_meta, load_npz, max_abs, rms, sub_sd

pytestmark = pytest.mark.gpu
DEV = "cuda"
KIND = {"conv": _lib.CONV_CAUSAL, "convt": _lib.CONV_TRANSPOSED, "upconv": _lib.CONV_UPSAMPLE}


def _load(module, sd):
    module.load_state_dict(sd)
    return module.to(DEV).eval()


# --------------------------------------------------------------------------- goldens
def test_golden_primitives_every_case():
    """Each primitive of tests/golden/g2 (the reference's own outputs) through the HIP path."""
    blob, cases = load_npz("g2_primitives.npz"), load_meta()["g2"]
    for c in cases:
        n = c["name"]
        sd = sub_sd(blob, f"{n}/sd/")
        x = torch.from_numpy(blob[f"{n}/x"]).to(DEV)
        if c["kind"] == "conv":
            m = CausalConv1d(c["cin"], c["cout"], c["k"], dilation=c["dilation"], stride=c["stride"])
        elif c["kind"] == "convt":
            m = CausalConvT1d(c["cin"], c["cout"], c["k"], stride=c["stride"])
        elif c["kind"] == "upconv":
            m = CausalUpsampleConv1d(c["cin"], c["cout"], c["k"], stride=c["stride"])
        elif c["kind"] == "res":
            m = CausalResidualBlock1d(c["c"], c["c"], dilation=c["dilation"])
        elif c["kind"] == "encblock":
            m = CausalEncoderBlock(c["cin"], c["cout"], c["stride"])
        elif c["kind"] == "decblock":
            m = CausalDecoderBlock(c["cin"], c["cout"], c["stride"])
        else:
            m = CausalDecoderBlock(c["cin"], c["cout"], c["stride"], upsample=False)
        with torch.no_grad():
            y = _load(m, sd)(x)
        want = blob[f"{n}/y"]
        assert tuple(y.shape) == want.shape, n
        assert max_abs(y.cpu(), want) < 5e-6, n


@pytest.mark.parametrize("fixture,key", [("g1_tiny_vqae.npz", "g1"), ("g5_om_wav.npz", "g1")])
def test_golden_tiny_vqae_stage_by_stage(fixture, key):
    g1, meta = load_npz("g1_tiny_vqae.npz"), load_meta()["g1"]["kwargs"]
    blob = load_npz(fixture)
    model = CausalVQAE(**{**meta, "strides": tuple(meta["strides"])})
    model.load_state_dict(sub_sd(g1, "sd/"), strict=False)
    model = model.to(DEV).eval()
    x = torch.from_numpy(blob["x"]).to(DEV)
    with torch.no_grad():
        h = model.encoders[0][1](x)
        stages = [h]
        for enc in list(model.encoders)[1:]:
            h = enc(h)
            stages.append(h)
        for i, s in enumerate(stages):
            if f"enc_stage_{i}" in blob:
                assert max_abs(s.cpu(), blob[f"enc_stage_{i}"]) < 5e-6, f"enc stage {i}"
        for i, dec in enumerate(model.decoders):
            h = dec(h)
            if f"dec_stage_{i}" in blob:
                assert max_abs(h.cpu(), blob[f"dec_stage_{i}"]) < 5e-6, f"dec stage {i}"
    assert max_abs(h.cpu(), blob["y"]) < 5e-6


# ------------------------------------------------------------ conv kernels vs oracle
SHAPES = [  # (kind, cin, cout, k, stride, dilation, B, L) -- the layer shapes of config S at short L
    ("conv", 1, 32, 7, 1, 1, 2, 700), ("conv", 32, 1, 7, 1, 1, 2, 700),
    ("conv", 32, 32, 7, 1, 1, 2, 1000), ("conv", 32, 32, 7, 1, 9, 2, 1000), ("conv", 32, 32, 1, 1, 1, 2, 700),
    ("conv", 32, 64, 5, 2, 1, 2, 1001), ("conv", 64, 64, 7, 1, 3, 2, 500), ("conv", 64, 128, 9, 4, 1, 2, 403),
    ("conv", 128, 128, 7, 1, 9, 1, 300), ("conv", 128, 256, 11, 5, 1, 2, 251), ("conv", 256, 256, 7, 1, 3, 1, 200),
    ("conv", 256, 512, 17, 8, 1, 1, 264), ("conv", 512, 512, 3, 1, 1, 2, 45), ("conv", 48, 40, 3, 1, 1, 1, 77),
    ("convt", 512, 512, 7, 1, 1, 2, 45), ("convt", 32, 16, 9, 4, 1, 1, 50),
    ("upconv", 512, 256, 17, 8, 1, 1, 45), ("upconv", 256, 128, 11, 5, 1, 1, 130), ("upconv", 128, 64, 9, 4, 1, 2, 200),
    ("upconv", 64, 32, 5, 2, 1, 2, 300), ("upconv", 16, 8, 4, 3, 1, 1, 33),
]


@pytest.mark.parametrize("impl", ["direct", "mfma", "bf16x3"])
def test_conv_kernels_match_oracle(impl):
    gen = torch.Generator().manual_seed(11)
    checked = 0
    for (kind, cin, cout, k, s, d, b, length) in SHAPES:
        if impl != "direct" and (cin % 16 != 0 or (cout * (s if kind != "conv" else 1)) < 32):
            continue
        wshape = (cin, cout, k) if kind == "convt" else (cout, cin, k)
        v = torch.randn(wshape, generator=gen) / (cin * k) ** 0.5
        g = torch.rand((wshape[0], 1, 1), generator=gen) + 0.5
        bias = torch.randn(cout, generator=gen) * 0.1
        x = torch.randn(b, cin, length, generator=gen)
        w = codec.fold_weight_norm(g, v)
        if kind == "conv":
            want = codec.causal_conv1d(x, w, bias, stride=s, dilation=d)
        elif kind == "convt":
            want = codec.causal_conv_t1d(x, w, bias, stride=s)
        else:
            want = codec.upsample_conv1d(x, w, bias, s)
        want = codec.leaky(want)
        desc = ops.conv_desc(KIND[kind], b, cin, cout, length, k, s, d, _lib.EPI_LEAKY_PRE, 0.1,
                             {"direct": _lib.IMPL_DIRECT, "mfma": _lib.IMPL_MFMA, "bf16x3": _lib.IMPL_MFMA_BF16X3}[impl])
        if impl == "bf16x3":
            assert ops.conv_kernel_name(desc).endswith(":bf16x3")
        packed = ops.conv_pack(desc, v.to(DEV), g.to(DEV))
        y = ops.conv_forward(desc, x.to(DEV), packed, bias.to(DEV))
        assert tuple(y.shape) == tuple(want.shape), (kind, cin, cout, k, s, d)
        err = max_abs(y.cpu(), want)
        assert err < 2e-5 * max(1.0, float(want.abs().max())), (impl, kind, cin, cout, k, s, d, err)
        checked += 1
    assert checked >= (len(SHAPES) if impl == "direct" else 14)


def test_residual_and_epilogues():
    gen = torch.Generator().manual_seed(5)
    for c, d, length in [(32, 1, 900), (32, 9, 515), (64, 3, 400), (128, 9, 300), (256, 1, 130), (8, 3, 100)]:
        spec_sd = {}
        for name, k in (("conv1", 7), ("conv2", 1)):
            v = torch.randn(c, c, k, generator=gen) / (c * k) ** 0.5
            spec_sd[f"{name}.conv.weight_v"] = v
            spec_sd[f"{name}.conv.weight_g"] = v.reshape(c, -1).norm(dim=1).reshape(-1, 1, 1) * 1.1
            spec_sd[f"{name}.conv.bias"] = torch.randn(c, generator=gen) * 0.1
        x = torch.randn(2, c, length, generator=gen)
        want = codec.residual_block(x, spec_sd, "", d)
        m = _load(CausalResidualBlock1d(c, c, dilation=d), spec_sd)
        with torch.no_grad():
            y_plain = m(x.to(DEV))
            y_act = m.run(x.to(DEV), 0.1)
        assert max_abs(y_plain.cpu(), want) < 3e-5, (c, d)
        assert max_abs(y_act.cpu(), codec.leaky(want)) < 3e-5, (c, d)
        # the two-launch form (conv + conv with fused epilogues) must agree with the single-kernel form
        try:
            CausalResidualBlock1d.split_launches = True
            with torch.no_grad():
                y_split = m.run(x.to(DEV), 0.1)
        finally:
            CausalResidualBlock1d.split_launches = False
        assert max_abs(y_split.cpu(), codec.leaky(want)) < 3e-5, (c, d)
        assert max_abs(y_split, y_act) < 3e-5, (c, d)
        # bf16x3 arithmetic (fused kernel for C in {32,64,128,256}): same tolerance as the fp32 kernels
        if c % 16 == 0 and c >= 32:
            for conv in (m.conv1, m.conv2):
                conv.impl = _lib.IMPL_MFMA_BF16X3
            try:
                with torch.no_grad():
                    y_bf = m.run(x.to(DEV), 0.1)
                    y_bf_plain = m(x.to(DEV))
            finally:
                for conv in (m.conv1, m.conv2):
                    conv.impl = _lib.IMPL_AUTO
            assert max_abs(y_bf.cpu(), codec.leaky(want)) < 3e-5, (c, d)
            assert max_abs(y_bf_plain.cpu(), want) < 3e-5, (c, d)


# ------------------------------------------------------------------------------- RVQ
def _rvq_case(b, t, d, k, q, seed, layout="b l c", dup=False, q_used=None, scale=1.0):
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(b, t, d, generator=gen) * scale
    cbs = torch.randn(q, k, d, generator=gen) * scale
    if dup:  # exact duplicates and exact hits: ties must go to the lowest index
        cbs[:, k // 2] = cbs[:, 1]
        x[0, 0] = cbs[0, 1]
        x[0, 1] = 0.5 * (cbs[0, 0] + cbs[0, 2])
    want_q, want_i, want_c = rvq.residual_quantize(x, cbs, q_used)
    xin = x.to(DEV)
    if layout == "b c l":
        xin = xin.transpose(1, 2).contiguous()
    qn = q if q_used is None else q_used
    xq, idx, sq, _ = ops.rvq_forward(xin, cbs.to(DEV), ops.rvq_pack(cbs.to(DEV)), qn, layout)
    if layout == "b c l":
        xq = xq.transpose(1, 2)
    assert idx.dtype == torch.int64 and tuple(idx.shape) == (b, t, qn)
    assert torch.equal(idx.cpu(), want_i), f"indices differ: {(idx.cpu() != want_i).sum().item()} of {want_i.numel()}"
    assert torch.equal(xq.cpu(), want_q), "x_q is not bit-identical"
    commit = float(sq.sum().item() / x.numel())
    assert abs(commit - float(want_c)) <= 1e-5 * max(1.0, abs(float(want_c)))


def test_rvq_bit_exact_default_shape():
    _rvq_case(4, 225, 512, 1024, 8, seed=1)


def test_rvq_bit_exact_channel_major_layout():
    _rvq_case(3, 75, 512, 1024, 8, seed=2, layout="b c l")


def test_rvq_bit_exact_ragged_and_small():
    _rvq_case(1, 50, 512, 1024, 1, seed=3)                  # config T: Q = 1, T = 50
    _rvq_case(2, 5, 16, 16, 1, seed=4)                      # tiny golden model's sizes
    _rvq_case(1, 37, 33, 100, 3, seed=5)                    # odd D, K not a multiple of 32, ragged frames
    _rvq_case(2, 31, 64, 300, 4, seed=6, layout="b c l")
    _rvq_case(1, 1, 8, 1, 2, seed=7)                        # single frame, single codeword


def test_rvq_large_codebook_and_wide_frames():
    """K > 2048: the |c'|^2 / |c'| tables stay in global memory (no LDS copy); D > 512: the sequential forms of
    the exact distance and of the update phase."""
    _rvq_case(1, 40, 64, 2304, 3, seed=21)
    _rvq_case(1, 33, 544, 96, 2, seed=22)


def test_rvq_one_codebook_size_per_stage():
    """The reference's per-quantizer ``codebook_size`` tuple (vae.py:233): stages of 1024 / 300 / 37 / 512 codewords in
    one (Q, 1024, D) tensor whose padding rows are poisoned -- they must never be selected, and the result must be the
    oracle's search over the real codewords only (bit-exact indices), also with duplicate codewords (ties -> the full
    defining search, which must stop at the stage's own size)."""
    torch.manual_seed(31)
    sizes = (1024, 300, 37, 512)
    q, k, d = len(sizes), 1024, 64
    cbs = torch.randn(q, k, d)
    for i, kq in enumerate(sizes):
        cbs[i] *= 0.7 ** i
        cbs[i, kq:] = 0.0            # what the module stores there; exactly the mean-ish region a careless search would pick
    cbs[2, :37] = cbs[2, :1].clone() + 1e-3 * torch.randn(37, d)      # a crowded stage: many candidates per frame
    cbs[2, 5] = cbs[2, 3]                                              # and an exact duplicate
    x = torch.randn(3, 70, d)
    want_q, want_i, want_c = rvq.residual_quantize(x, cbs, sizes=sizes)
    packed = ops.rvq_pack(cbs.to(DEV), sizes)
    xq, idx, sq, _ = ops.rvq_forward(x.to(DEV), cbs.to(DEV), packed, q)
    assert torch.equal(idx.cpu(), want_i)
    for i, kq in enumerate(sizes):
        assert int(idx[..., i].max()) < kq
    assert torch.equal(xq.cpu(), want_q)
    from audio_generation_amd.quantizer import ResidualQuantizer
    mod = ResidualQuantizer(num_quantizers=q, dim=d, codebook_sizes=sizes).to(DEV).eval()
    assert mod.codebook_sizes == sizes and mod.codebooks.shape == (q, 1024, d)
    assert [tuple(s.codebook.shape) for s in mod.quantizers] == [(kq, d) for kq in sizes]
    with torch.no_grad():
        mod.codebooks.copy_(cbs.to(DEV))          # through the tensor itself: bumps _version -> the search image is repacked
    got_q, got_i, _ = mod(x.to(DEV))
    assert torch.equal(got_i.cpu(), want_i) and torch.equal(got_q.cpu(), want_q)


def test_rvq_ties_duplicates_and_truncation():
    _rvq_case(2, 40, 64, 128, 4, seed=8, dup=True)
    _rvq_case(2, 40, 64, 128, 4, seed=9, q_used=2)
    _rvq_case(1, 20, 32, 64, 3, seed=10, q_used=0)
    # degenerate codebook: every codeword identical -> candidate overflow -> full defining search
    b, t, d, k = 1, 33, 32, 64
    x = torch.randn(b, t, d)
    cbs = torch.randn(1, 1, d).expand(1, k, d).contiguous()
    _, want_i, _ = rvq.residual_quantize(x, cbs)
    _, idx, _, _ = ops.rvq_forward(x.to(DEV), cbs.to(DEV), ops.rvq_pack(cbs.to(DEV)), 1)
    assert torch.equal(idx.cpu(), want_i) and int(idx.max()) == 0


def test_rvq_small_scale_latents():
    # encoder outputs at random init are ~1e-2: margins must scale with the data
    _rvq_case(2, 100, 512, 1024, 8, seed=12, scale=0.02)


def test_rvq_latents_with_large_common_offset():
    """Encoder outputs share a big bias component; codewords close to actual frames.  The score
    kernel centres on the stage's mean codeword so its fp32 margin stays tight: still bit-exact,
    and the single-launch kernel must not fall into the full fp64 search for every frame."""
    gen = torch.Generator().manual_seed(21)
    b, t, d, k, q = 4, 225, 512, 1024, 4
    mean = 3.0 * torch.randn(d, generator=gen)
    x = mean + 0.02 * torch.randn(b, t, d, generator=gen)
    frames = x.reshape(-1, d)
    cbs = torch.randn(q, k, d, generator=gen) * 0.01
    cbs[0] = frames[torch.randint(0, frames.shape[0], (k,), generator=gen)] + 0.002 * torch.randn(k, d, generator=gen)
    want_q, want_i, _ = rvq.residual_quantize(x, cbs)
    xd, cd = x.to(DEV), cbs.to(DEV)
    packed = ops.rvq_pack(cd)
    xq, idx, _, _ = ops.rvq_forward(xd, cd, packed, q)
    assert torch.equal(idx.cpu(), want_i) and torch.equal(xq.cpu(), want_q)
    assert want_i[..., 0].unique().numel() > 200
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        ops.rvq_forward(xd, cd, packed, q)
    e1.record()
    torch.cuda.synchronize()
    assert e0.elapsed_time(e1) / 3 < 2.0, "RVQ fell off the fast path (ms per launch)"


def test_dequantize_gather():
    cb = torch.randn(50, 24)
    idx = torch.randint(0, 50, (3, 7))
    out = ops.rvq_dequantize(cb.to(DEV), idx.to(DEV))
    assert torch.equal(out.cpu(), cb[idx])


# ------------------------------------------------------------------------- end to end
def _oracle_forward_from_gpu_latents(model, x, z_gpu):
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    return sd, rvq.residual_quantize(z_gpu.cpu().transpose(1, 2).contiguous(), sd["quantizer.codebooks"])


@pytest.mark.parametrize("fmt", ["n c l", "b l c"])
def test_end_to_end_soundstream_default(fmt):
    """Config S topology (8 x 1024 x 512 RVQ, strides 2,4,5,8, 32..512 channels) at B=2, L=9600."""
    torch.manual_seed(0)
    model = CausalVQAE(in_channels=1, n_blocks=4, strides=(2, 4, 5, 8), num_quantizers=8, codebook_size=1024,
                       codebook_dim=512, input_format=fmt, wavelet_decoders=False).eval()
    gen = torch.Generator().manual_seed(1234)
    x = (0.1 * torch.randn(2, 1, 9600, generator=gen)).clamp(-1, 1)
    if fmt == "b l c":
        x = x.transpose(1, 2).contiguous()
    spec = codec.CodecSpec(in_channels=1, n_blocks=4, strides=(2, 4, 5, 8), codebook_dim=512,
                           wavelet_decoders=False, input_format=fmt)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    # codebooks at the scale of the latents so that the arg-min is non-degenerate
    z_ref = codec.encode_latents(x, sd, spec)
    sigma = model.quantizer.init_from_latents(z_ref.transpose(1, 2))
    sd["quantizer.codebooks"] = model.quantizer.codebooks.detach().clone()
    model = model.to(DEV)
    with torch.no_grad():
        y, commit, index = model(x.to(DEV))
        z_gpu = model._run_encoders(model.rearrange_in(x.to(DEV)))
    assert tuple(index.shape) == (2, 30, 8) and y.shape == x.shape
    assert index[..., 0].unique().numel() > 20      # the arg-min is exercised, not one code for all frames
    # (1) encoder latents within fp32 rounding of the oracle's
    assert rms(z_gpu.cpu().transpose(1, 2), z_ref) < 1e-5 * max(1.0, sigma)
    # (2) indices bit-exact against the oracle run on the SAME latents
    zq_o, idx_o, commit_o = rvq.residual_quantize(z_gpu.cpu().transpose(1, 2).contiguous(), sd["quantizer.codebooks"])
    assert torch.equal(index.cpu(), idx_o)
    assert abs(float(commit) - float(commit_o)) < 1e-5 * max(1.0, float(commit_o))
    # (3) waveform within the 1e-4 RMS budget of the oracle decode of the same codes
    y_o = codec.decode_latents(zq_o, sd, spec)
    assert rms(y.cpu(), y_o) < 1e-4
    # (4) fully independent CPU path: report agreement; any flip must be a near tie
    zq_i, idx_i, _ = rvq.residual_quantize(z_ref, sd["quantizer.codebooks"])
    agree = float((idx_i == index.cpu()).float().mean())
    print(f"[{fmt}] independent-path index agreement {agree:.4f}; waveform RMS vs oracle "
          f"{rms(y.cpu(), codec.decode_latents(zq_i, sd, spec)):.3e}")
    # ... proved, not assumed: both top-2 margins of every first disagreement are below 2 |delta| |c_a - c_b| with
    # delta the measured latent difference of that frame (oracle/neartie.py)
    rep = neartie.explain_disagreements(z_gpu.cpu().transpose(1, 2).reshape(-1, 512).numpy(), z_ref.reshape(-1, 512).numpy(),
                                        index.cpu().reshape(-1, 8).numpy(), idx_i.reshape(-1, 8).numpy(),
                                        sd["quantizer.codebooks"].numpy())
    assert rep["proved"] and rep["max_latent_error_relative"] < 2e-5, rep


def test_sample_and_codebook_n():
    torch.manual_seed(3)
    model = CausalVQAE(in_channels=1, n_blocks=4, strides=(2, 4, 5, 8), first_block_channels=8, num_quantizers=4,
                       codebook_size=32, codebook_dim=64, input_format="n c l", wavelet_decoders=False).to(DEV).eval()
    y = model.sample(length=10, device=DEV)
    assert tuple(y.shape) == (1, 1, 3200) and torch.isfinite(y).all()
    x = 0.1 * torch.randn(1, 1, 3200, device=DEV)
    with torch.no_grad():
        _, _, idx2 = model(x, codebook_n=2)
        _, _, idx4 = model(x)
    assert tuple(idx2.shape) == (1, 10, 2) and torch.equal(idx2, idx4[..., :2])


def test_ragged_length_matches_reference_padding():
    """L not a multiple of the total stride: _calc_extra_pad (vae.py:39-43) right-pads every strided conv."""
    torch.manual_seed(4)
    kw = dict(in_channels=1, n_blocks=4, strides=(2, 4, 5, 8), first_block_channels=8, num_quantizers=2,
              codebook_size=32, codebook_dim=32, input_format="n c l", wavelet_decoders=False)
    model = CausalVQAE(**kw).eval()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    spec = codec.CodecSpec(in_channels=1, n_blocks=4, strides=(2, 4, 5, 8), first_block_channels=8,
                           codebook_dim=32, wavelet_decoders=False, input_format="n c l")
    x = 0.1 * torch.randn(2, 1, 1234)
    z_ref = codec.encode_latents(x, sd, spec)
    model = model.to(DEV)
    with torch.no_grad():
        z = model._run_encoders(x.to(DEV))
        y = model.decode(z)
    assert tuple(z.shape) == (2, 32, z_ref.shape[1])
    assert max_abs(z.cpu().transpose(1, 2), z_ref) < 1e-5
    assert max_abs(y.cpu(), codec.decode_latents(z.cpu().transpose(1, 2), sd, spec)) < 1e-5


# ------------------------------------------------------------------------- bitstream
@pytest.mark.parametrize("bits,n", [(10, 32 * 225 * 8), (10, 7), (9, 1001), (1, 64), (16, 33), (12, 8)])
def test_code_packing_bit_exact_and_round_trip(bits, n):
    from oracle import bitstream
    gen = torch.Generator().manual_seed(bits * 1000 + n)
    codes = torch.randint(0, 2 ** bits, (n,), generator=gen)
    packed = ops.codes_pack(codes.to(DEV), bits)
    want = bitstream.pack(codes.numpy(), bits)
    assert packed.dtype == torch.uint8 and packed.numel() == (n * bits + 7) // 8 == want.size
    assert np.array_equal(packed.cpu().numpy(), want)
    back = ops.codes_unpack(packed, n, bits)
    assert torch.equal(back.cpu(), codes)
    assert np.array_equal(bitstream.unpack(want, n, bits), codes.numpy())


def test_shipped_yaml_config_topology():
    """config/training.yml `vae_args`: default 5 blocks, strides (2,3,4,4,5), channels 32..1024, wavelet
    layer in the second decoder block, 10 x 512 'base' (learnable) codebooks, 'n c l' input."""
    torch.manual_seed(6)
    model = CausalVQAE(in_channels=1, num_quantizers=10, codebook_size=512, input_format="n c l", vq_cutoff_freq=0.1,
                       use_som=True, som_kernel_type="hard", vq_type="base").eval()
    assert model.scale_factor == 480 and len(list(model.quantizer.parameters())) == 1
    spec = codec.CodecSpec(in_channels=1, input_format="n c l")          # every default of vae.py:205-223
    x = 0.1 * torch.randn(1, 1, 4800)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    z_ref = codec.encode_latents(x, sd, spec)
    with torch.no_grad():
        model.quantizer.init_from_latents(z_ref.transpose(1, 2))
    sd["quantizer.codebooks"] = model.quantizer.codebooks.detach().clone()
    model = model.to(DEV)
    with torch.no_grad():
        y, commit, index = model(x.to(DEV))
        z = model._run_encoders(x.to(DEV))
    assert tuple(index.shape) == (1, 10, 10) and tuple(z.shape) == (1, 512, 10)
    assert rms(z.cpu().transpose(1, 2), z_ref) < 1e-5
    zq_o, idx_o, _ = rvq.residual_quantize(z.cpu().transpose(1, 2).contiguous(), sd["quantizer.codebooks"])
    assert torch.equal(index.cpu(), idx_o)
    assert rms(y.cpu(), codec.decode_latents(zq_o, sd, spec)) < 1e-4
