"""Size-independent properties at BASELINE.json's full sizes (config S: batch 32 x 72 000
samples, 8 x 1024 x 512 RVQ, strides 2,4,5,8, fp32) -- where the CPU oracle would take
minutes, the domain's own invariants are checked instead (exactly, where fp32 allows)."""
import pytest
import torch

from audio_generation_amd import ops
from audio_generation_amd.graph import GraphedForward
from audio_generation_amd.vae import CausalVQAE

pytestmark = pytest.mark.gpu
DEV = "cuda"
B, L = 32, 72000


@pytest.fixture(scope="module")
def setup():
    torch.manual_seed(0)
    model = CausalVQAE(in_channels=1, n_blocks=4, strides=(2, 4, 5, 8), num_quantizers=8, codebook_size=1024,
                       codebook_dim=512, input_format="n c l", wavelet_decoders=False).to(DEV).eval()
    gen = torch.Generator().manual_seed(1234)
    x = (0.1 * torch.randn(B, 1, L, generator=gen)).clamp(-1, 1).to(DEV)
    with torch.no_grad():
        model.quantizer.init_from_latents(model._run_encoders(x[:8]))
        y, commit, index = model(x)
        z = model._run_encoders(x)
    return model, x, z, y, commit, index


def test_shapes_and_known_answers(setup):
    model, x, z, y, commit, index = setup
    assert tuple(z.shape) == (B, 512, 225)            # vae.py:354: 72000 -> 225 frames
    assert tuple(index.shape) == (B, 225, 8) and index.dtype == torch.int64   # utils.py:249
    assert y.shape == x.shape and torch.isfinite(y).all() and torch.isfinite(commit)
    assert int(index.min()) >= 0 and int(index.max()) < 1024
    # every stage uses a healthy part of its codebook (arg-min is not degenerate)
    for q in range(8):
        assert index[..., q].unique().numel() > 64, (q, index[..., q].unique().numel())


def test_batch_items_are_independent_and_order_free(setup):
    """Sharding invariant behind the multi-GPU path: permuting / slicing the batch permutes /
    slices every output bit for bit (no cross-item op, tiling does not depend on the batch index)."""
    model, x, z, y, commit, index = setup
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(3)).to(DEV)
    with torch.no_grad():
        y_p, _, idx_p = model(x[perm])
        y_s, _, idx_s = model(x[5:9])
    assert torch.equal(y_p, y[perm]) and torch.equal(idx_p, index[perm])
    assert torch.equal(y_s, y[5:9]) and torch.equal(idx_s, index[5:9])


def test_encoder_is_causal_bit_for_bit(setup):
    """A prefix of the waveform gives exactly the prefix of the latents (vae.py:14-43: every
    encoder conv is causal), whatever the time tiling of the kernels."""
    model, x, z, *_ = setup
    cut = 320 * 77                                      # 77 frames
    with torch.no_grad():
        z_prefix = model._run_encoders(x[:4, :, :cut].contiguous())
        x_mod = x[:4].clone()
        x_mod[:, :, cut:] = 0.5                          # change the future only
        z_mod = model._run_encoders(x_mod)
    assert torch.equal(z_prefix, z[:4, :, :77])
    assert torch.equal(z_mod[:, :, :77], z[:4, :, :77]) and not torch.equal(z_mod, z[:4])


def test_rvq_residual_identities(setup):
    model, x, z, y, commit, index = setup
    cb = model.quantizer.codebooks
    # x_q is the sum of the selected codewords in stage order (bit-exact re-assembly by gather)
    with torch.no_grad():
        zq, idx, sq, _ = ops.rvq_forward(z, cb, ops.rvq_pack(cb), 8, "b c l")
        acc = None
        for q in range(8):
            acc = ops.rvq_dequantize(cb[q], idx[..., q]) if acc is None else \
                ops.rvq_dequantize(cb[q], idx[..., q], out=acc, accumulate=True)
    assert torch.equal(idx, index)
    assert torch.equal(acc.transpose(1, 2), zq)
    # the squared residual reported for the last stage is ||z - x_q||^2
    r = z.double() - zq.double()
    assert abs(float(sq[7]) - float((r * r).sum())) < 1e-5 * float(sq[7]) + 1e-9
    # truncating the stage loop is a prefix of the full result
    _, idx3, _, _ = ops.rvq_forward(z, cb, ops.rvq_pack(cb), 3, "b c l")
    assert torch.equal(idx3, index[..., :3])
    # nearest-neighbour optimality spot check in float64 on 2 000 random (frame, stage-0) pairs
    frames = z.transpose(1, 2).reshape(-1, 512)
    pick = torch.randint(0, frames.shape[0], (2000,), generator=torch.Generator().manual_seed(5)).to(DEV)
    d = torch.cdist(frames[pick].double(), cb[0].double())
    assert torch.equal(d.argmin(dim=1), index.reshape(-1, 8)[pick, 0])


def test_decoder_linearity_and_determinism(setup):
    model, x, z, y, commit, index = setup
    with torch.no_grad():
        y2, _, idx2 = model(x)
    assert torch.equal(y2, y) and torch.equal(idx2, index)      # run-to-run bit reproducible
    # a single conv layer is linear: scaling the input by 2 (exact in fp32) scales the bias-free part by 2
    conv = model.encoders[1].layers[3][0]                        # strided 32 -> 64 down conv
    h = torch.randn(B, 32, 4000, device=DEV)
    with torch.no_grad():
        zero = conv.run(torch.zeros_like(h))
        a, b2 = conv.run(h), conv.run(2 * h)
    assert torch.equal(b2 - zero, 2 * (a - zero)) or (b2 - zero - 2 * (a - zero)).abs().max() < 1e-5


def test_graph_replay_matches_eager(setup):
    model, x, z, y, commit, index = setup
    g = GraphedForward(model, x[:8])
    yg, cg, ig = g.replay()
    assert torch.equal(yg, y[:8]) and torch.equal(ig, index[:8])
    yg2, _, ig2 = g(x[8:16])
    assert torch.equal(yg2, y[8:16]) and torch.equal(ig2, index[8:16])


def test_bitstream_round_trip_reproduces_the_forward(setup):
    """compress -> 10-bit stream -> decompress gives exactly the codes and waveform of forward()."""
    model, x, z, y, commit, index = setup
    stream, shape = model.compress(x[:4])
    assert stream.numel() == (4 * 225 * 8 * 10 + 7) // 8 and shape == (4, 225, 8)    # 9 000 bytes for 12 s of audio
    y2, idx2 = model.decompress(stream, shape)
    assert torch.equal(idx2, index[:4]) and torch.equal(y2, y[:4])


def test_bf16x3_decoder_matches_fp32_and_oracle(setup):
    """Decoder on the bf16x3 kernels (CausalVQAE.set_conv_arithmetic): same waveform as the fp32-MFMA decoder
    to ~1e-6 of its scale, and within the 1e-4 RMS budget of the oracle; indices unchanged (encoder stays fp32)."""
    model, x, z, y, commit, index = setup
    model.set_conv_arithmetic(decoders="bf16x3")
    try:
        with torch.no_grad():
            y2, _, index2 = model(x)
    finally:
        model.set_conv_arithmetic()
    assert torch.equal(index2, index)
    scale = float(y.abs().max())
    assert float((y2 - y).abs().max()) <= 2e-5 * scale
    assert float((y2 - y).pow(2).mean().sqrt()) <= 1e-6 * max(scale, 1.0)


def test_bf16x3_decoder_head_on_activation_planes_is_bit_identical(setup):
    """Round 4: with the decoder on the bf16x3 kernels the k = 7 transposed conv writes ACTIVATION PLANES and the first
    up-conv stages them by LDS-DMA (CausalVQAE._decoder_head_on_planes).  Same pieces, same products, same order: the
    waveform must equal the fp32-activation path bit for bit -- at the full size of config S and on a ragged short clip."""
    model, x, z, y, commit, index = setup
    model.set_conv_arithmetic(decoders="bf16x3")
    calls = {"n": 0}
    real = ops.conv_forward_planes

    def counting(*a, **k):
        calls["n"] += 1
        return real(*a, **k)

    try:
        ops.conv_forward_planes = counting
        with torch.no_grad():
            y_planes, _, idx_planes = model(x)
            y_short, _, _ = model(x[:3, :, :320 * 37])
        ops.conv_forward_planes = real
        assert calls["n"] == 4                                  # (k7 + up8) x two forwards went through the planes path
        head = type(model)._decoder_head_on_planes
        type(model)._decoder_head_on_planes = lambda self, h, decs: (h, decs)     # the plain path
        try:
            with torch.no_grad():
                y_plain, _, idx_plain = model(x)
                y_short_plain, _, _ = model(x[:3, :, :320 * 37])
        finally:
            type(model)._decoder_head_on_planes = head
    finally:
        ops.conv_forward_planes = real
        model.set_conv_arithmetic()
    assert torch.equal(idx_planes, idx_plain) and torch.equal(idx_planes, index)
    assert torch.equal(y_planes, y_plain) and torch.equal(y_short, y_short_plain)
