"""Discriminators on the HIP kernels against the oracle and the reference's goldens (G7)."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from audio_generation_amd import discriminator as ad
from audio_generation_amd import ops
from audio_generation_amd._lib import CONV_PADDED, EPI_LEAKY_PRE, IMPL_DIRECT, IMPL_MFMA
from oracle import discriminator as od

pytestmark = pytest.mark.gpu
DEV = "cuda"
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
    a, b = a.detach().cpu(), b.detach().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = float(b.abs().max()) + 1e-12
    err = float((a - b).abs().max())
    assert err <= tol * scale + 1e-7, err / scale


# ------------------------------------------------------------------ primitives
@pytest.mark.parametrize("win", [64, 256, 1024])
def test_stft_matches_torch_stft_golden(g7, win):
    x, want = g7[f"stft_only_{win}/x"], g7[f"stft_only_{win}/y"]            # want (B, F, T, 2)
    got = ops.stft(x.to(DEV), win, True)
    close(got, want.permute(0, 3, 2, 1), 2e-5)


def test_stft_full_size_against_oracle():
    torch.manual_seed(0)
    x = 0.3 * torch.randn(3, 72000)
    for win in (2048, 1024, 512):
        got = ops.stft(x.to(DEV), win, True)
        assert got.shape == (3, 2, 1 + 72000 // (win // 4), win)
        close(got, od.stft_two_sided(x, win, win // 4), 2e-5)
    # unnormalised, and the Hermitian symmetry of a real signal's two-sided spectrum: X[N - f] = conj X[f]
    got = ops.stft(x.to(DEV), 256, False).cpu()
    close(got, od.stft_two_sided(x, 256, 64, False), 2e-5)
    close(got[:, 0, :, 1:], got[:, 0, :, 1:].flip(-1), 1e-4)
    close(got[:, 1, :, 1:], -got[:, 1, :, 1:].flip(-1), 1e-4)


@pytest.mark.parametrize("cin,cout,k,s,g,pad,length,impl", [
    (1, 16, 15, 1, 1, 0, 700, 0), (16, 64, 41, 4, 4, 0, 3000, 0), (64, 256, 41, 4, 16, 0, 900, 0),
    (256, 512, 41, 4, 64, 0, 300, 0), (512, 1024, 41, 4, 256, 0, 200, 0), (64, 64, 5, 1, 1, 0, 77, IMPL_MFMA),
    (64, 64, 5, 1, 1, 0, 77, IMPL_DIRECT), (32, 1, 3, 1, 1, 0, 19, 0), (6, 9, 4, 3, 3, 2, 50, 0),
    (16, 32, 7, 2, 1, 5, 333, IMPL_MFMA)])
def test_padded_grouped_conv1d(cin, cout, k, s, g, pad, length, impl):
    torch.manual_seed(cin + cout)
    x = torch.randn(2, cin, length)
    w = torch.randn(cout, cin // g, k) / (cin // g * k) ** 0.5
    b = torch.randn(cout)
    want = F.leaky_relu(F.conv1d(x, w, b, stride=s, padding=pad, groups=g), 0.2)
    d = ops.conv_desc(CONV_PADDED, 2, cin, cout, length, k, s, 1, EPI_LEAKY_PRE, 0.2, impl, groups=g, padding=pad)
    got = ops.conv_forward(d, x.to(DEV), ops.conv_pack(d, w.to(DEV)), b.to(DEV))
    close(got, want, 1e-5)


@pytest.mark.parametrize("scale,length", [(1, 100), (2, 101), (4, 1000)])
def test_avgpool1d(scale, length):
    x = torch.randn(3, 2, length)
    close(ops.avgpool1d(x.to(DEV), 2 * scale, scale, scale), F.avg_pool1d(x, 2 * scale, stride=scale, padding=scale), 1e-6)


@pytest.mark.parametrize("shape", [(16, 1, 15), (64, 4, 41), (32, 32, 3, 3), (1, 512, 1, 8), (4, 2, 7, 7)])
@pytest.mark.parametrize("n_iter", [0, 1, 3])
def test_spectral_sigma(shape, n_iter):
    torch.manual_seed(len(shape) + n_iter)
    w = torch.randn(*shape)
    rows, cols = shape[0], w.numel() // shape[0]
    u, v = F.normalize(torch.randn(rows), dim=0), F.normalize(torch.randn(cols), dim=0)
    sd = {"weight_orig": w, "weight_u": u.clone(), "weight_v": v.clone()}
    for _ in range(max(n_iter, 1)):
        wn = od.spectral_weight(sd, "", train=n_iter > 0)
    sigma_want = float((w / wn).flatten()[0])
    ud, vd = u.to(DEV), v.to(DEV)
    sigma = ops.spectral_sigma(w.to(DEV), ud, vd, n_iter)
    assert abs(float(sigma) - sigma_want) <= 2e-5 * abs(sigma_want) + 1e-7
    close(ud, sd["weight_u"], 2e-5)
    close(vd, sd["weight_v"], 2e-5)


@pytest.mark.parametrize("cin,cout,kh,kw,sh,sw,ph,pw,h,w,impl", [
    (2, 32, 7, 7, 1, 1, 3, 3, 21, 70, 0), (2, 8, 7, 7, 1, 1, 3, 3, 9, 64, IMPL_DIRECT),
    (32, 32, 3, 3, 1, 1, 1, 1, 17, 200, 0), (32, 64, 3, 4, 1, 2, 1, 1, 17, 200, 0),
    (64, 128, 4, 4, 2, 2, 1, 1, 18, 130, 0), (128, 128, 3, 3, 1, 1, 1, 1, 9, 64, 0),
    (16, 32, 3, 3, 1, 1, 1, 1, 5, 33, IMPL_MFMA), (16, 32, 3, 3, 1, 1, 1, 1, 5, 33, IMPL_DIRECT),
    (512, 1, 1, 8, 1, 1, 0, 3, 5, 16, 0), (4, 4, 3, 3, 1, 1, 1, 1, 1, 5, 0), (48, 40, 2, 5, 3, 2, 0, 4, 11, 41, 0)])
def test_conv2d(cin, cout, kh, kw, sh, sw, ph, pw, h, w, impl):
    torch.manual_seed(cin * 7 + kh)
    x = torch.randn(2, cin, h, w)
    wt = torch.randn(cout, cin, kh, kw) / (cin * kh * kw) ** 0.5
    b = torch.randn(cout)
    want = F.leaky_relu(F.conv2d(x, wt, b, stride=(sh, sw), padding=(ph, pw)), 0.2)
    d = ops.conv2d_desc(2, cin, cout, h, w, kh, kw, (sh, sw), (ph, pw), EPI_LEAKY_PRE, 0.2, impl)
    got = ops.conv2d_forward(d, x.to(DEV), ops.conv2d_pack(d, wt.to(DEV)), b.to(DEV))
    close(got, want, 1e-5)
    if impl == 0:   # the bf16x3 arithmetic (layers without a bf16x3 form fall back to fp32 inside the library)
        from audio_generation_amd._lib import IMPL_MFMA_BF16X3
        d3 = ops.conv2d_desc(2, cin, cout, h, w, kh, kw, (sh, sw), (ph, pw), EPI_LEAKY_PRE, 0.2, IMPL_MFMA_BF16X3)
        close(ops.conv2d_forward(d3, x.to(DEV), ops.conv2d_pack(d3, wt.to(DEV)), b.to(DEV)), want, 1e-5)
    name = ops.conv2d_kernel_name(d)
    if impl == IMPL_MFMA or (impl == 0 and cout >= 32):
        assert name.startswith(("conv_mfma", "conv_p2d")), name


@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4])
def test_loss_means_and_their_gradients(mode):
    torch.manual_seed(mode)
    x = (1.5 * torch.randn(3, 5, 1000)).requires_grad_(True)
    y = (1.5 * torch.randn(3, 5, 1000)).requires_grad_(True)
    want = [lambda: x.mean(), lambda: torch.minimum(x - 1, torch.zeros_like(x)).mean(),
            lambda: torch.minimum(-x - 1, torch.zeros_like(x)).mean(), lambda: F.l1_loss(x, y),
            lambda: torch.abs(x + 1e-3).mean()][mode]()
    (want * 3.0).backward()
    xd, yd = x.detach().to(DEV).requires_grad_(True), y.detach().to(DEV).requires_grad_(True)
    got = ad._mean(mode, xd, yd if mode == 3 else None)
    assert abs(float(got) - float(want)) <= 1e-6 * abs(float(want)) + 1e-7
    (got * 3.0).backward()
    close(xd.grad, x.grad, 1e-6)
    if mode == 3:
        close(yd.grad, y.grad, 1e-6)


# ------------------------------------------------------------------ modules vs the reference's goldens
def _wave_block(meta, scale):
    kw = meta["wave"]["kwargs"]
    return ad.WaveformDiscriminatorBlock(1, channel_sizes=kw["channel_sizes"], groups=kw["groups"], scale=scale)


@pytest.mark.parametrize("scale", [1, 2])
def test_waveform_block_matches_reference(g7, meta, scale):
    blk = _wave_block(meta, scale)
    name = f"wave_s{scale}_eval"
    blk.load_state_dict(sub(g7, name + "/sd/"), strict=True)
    blk = blk.to(DEV).eval()
    with torch.no_grad():
        out, feats = blk(g7[name + "/x"].to(DEV))
    # eval mode on never-iterated u / v: sigma is a small random number, the activations are huge and
    # sigmoid saturates (the reference does exactly this) -- compare the features relative to their scale
    assert len(feats) == meta["wave"]["n_features"]
    for i, f in enumerate(feats):
        close(f, g7[f"{name}/feat{i}"], 2e-4)
    close(out, g7[name + "/out0"], 1e-5)
    # training mode: one power iteration per forward, buffers move exactly as the reference's
    name = f"wave_s{scale}_train"
    blk.load_state_dict(sub(g7, name + "/sd/"), strict=True)
    blk.train()
    with torch.no_grad():
        out, _ = blk(g7[name + "/x"].to(DEV))
    close(out, g7[name + "/out0"], 2e-5)
    sd = blk.state_dict()
    for k, v in sub(g7, name + "/sd_after/").items():
        close(sd[k], v, 2e-5)


def test_state_dict_keys_match_reference(meta):
    assert list(ad.WaveformDiscriminatorBlock(1).state_dict().keys()) == meta["wave_default_keys"]
    kw = meta["stft"]["kwargs"]
    d = ad.STFTDiscriminator(**kw)
    assert list(d.state_dict().keys()) == meta["stft"]["keys"] and d.name == meta["stft"]["name"]
    # default layer tables
    full = ad.STFTDiscriminator()
    rows = [m for m in full.modules() if isinstance(m, ad._SNConv)]
    assert [(m.in_channels, m.out_channels, list(m.kernel_size), list(m.stride), list(m.padding)) for m in rows] == \
           [(r["cin"], r["cout"], r["k"], r["stride"], r["padding"]) for r in meta["stft_default_layers"]]
    rows = [m for m in ad.WaveformDiscriminatorBlock(1).modules() if isinstance(m, ad._SNConv)]
    assert [(m.in_channels, m.out_channels, m.kernel_size[0], m.stride[0], m.groups) for m in rows] == \
           [(r["cin"], r["cout"], r["k"], r["stride"], r["groups"]) for r in meta["wave_default_layers"]]


def test_stft_discriminator_matches_reference(g7, meta):
    kw = meta["stft"]["kwargs"]
    d = ad.STFTDiscriminator(**kw)
    d.load_state_dict(sub(g7, "stft_eval/sd/"), strict=True)
    d = d.to(DEV).eval()
    with torch.no_grad():
        outs, feats = d(g7["stft_eval/x"].to(DEV))
    assert len(feats) == meta["stft"]["n_features"] and len(outs) == 1
    for i, f in enumerate(feats):
        close(f, g7[f"stft_eval/feat{i}"], 5e-4)
    close(outs[0], g7["stft_eval/out0"], 1e-4)
    d.load_state_dict(sub(g7, "stft_train/sd/"), strict=True)
    d.train()
    with torch.no_grad():
        outs, _ = d(g7["stft_train/x"].to(DEV))
    close(outs[0], g7["stft_train/out0"], 5e-5)
    sd = d.state_dict()
    for k, v in sub(g7, "stft_train/sd_after/").items():
        close(sd[k], v, 2e-5)


@pytest.mark.parametrize("tag", ["wave", "stft"])
def test_loss_matches_reference_and_oracle_gradients(g7, meta, tag):
    orig, rec = g7["loss/original"], g7["loss/reconstruction"]
    sd0 = sub(g7, f"loss_{tag}/sd/")
    if tag == "wave":
        disc = ad.WaveFormDiscriminator(1, n_blocks=2)
        disc.layers = torch.nn.ModuleList([_wave_block(meta, s) for s in (1, 2)])
        groups = meta["wave"]["kwargs"]["groups"]
        oracle = lambda sd: (lambda t: od.waveform_discriminator(t, sd, n_blocks=2, train=True, groups=groups))  # noqa: E731
    else:
        disc = ad.STFTDiscriminator(**meta["stft"]["kwargs"])
        win = meta["stft"]["kwargs"]["win_length"]
        oracle = lambda sd: (lambda t: od.stft_discriminator(t, sd, win, train=True))  # noqa: E731
    disc.load_state_dict(sd0, strict=True)
    disc = disc.to(DEV).train()
    rec_d = rec.to(DEV).requires_grad_(True)
    gl, dl = ad.discriminator_generator_loss(orig.to(DEV), rec_d, disc)
    close(gl, g7[f"loss_{tag}/generator_loss"], 2e-4)
    close(dl, g7[f"loss_{tag}/discriminator_loss"], 2e-5)
    # gradients of both losses against the oracle's autograd (same state, same three passes)
    sd = {k: (v.clone().requires_grad_(True) if k.endswith("weight_orig") or k.endswith("bias") else v.clone())
          for k, v in sd0.items()}
    rec_c = rec.clone().requires_grad_(True)
    gl_o, dl_o = od.discriminator_generator_loss(orig, rec_c, oracle(sd))
    g_rec = torch.autograd.grad(gl_o, rec_c, retain_graph=True)[0]
    leaves = [k for k in sd if sd[k].requires_grad]
    g_par = torch.autograd.grad(dl_o, [sd[k] for k in leaves], allow_unused=True)
    got_rec = torch.autograd.grad(gl, rec_d, retain_graph=True)[0]
    close(got_rec, g_rec, 2e-3)
    params = dict(disc.named_parameters())
    got_par = torch.autograd.grad(dl, [params[k] for k in leaves], allow_unused=True)
    checked = 0
    for k, a, b in zip(leaves, got_par, g_par):
        if b is None:
            assert a is None or float(a.abs().max()) == 0.0
            continue
        close(a, b, 3e-3)
        checked += 1
    assert checked >= 14
    # second call continues from the moved buffers, like the reference's second call
    gl2, dl2 = ad.discriminator_generator_loss(orig.to(DEV), rec.to(DEV), disc, feature_multipier=3.0,
                                               scale_feature_loss=False)
    close(gl2, g7[f"loss_{tag}/generator_loss_unscaled_fm3"], 2e-4)
    close(dl2, g7[f"loss_{tag}/discriminator_loss_2nd_call"], 2e-5)


def test_full_size_stft_discriminator_against_oracle():
    """Default widths, win 1024, one 24 kHz second: every feature map against the CPU restatement."""
    torch.manual_seed(3)
    d = ad.STFTDiscriminator()
    sd = {k: v.clone() for k, v in d.state_dict().items()}
    x = 0.3 * torch.randn(2, 1, 24000)
    d = d.to(DEV).train()
    with torch.no_grad():
        outs, feats = d(x.to(DEV))
    want_o, want_f = od.stft_discriminator(x, sd, 1024, train=True)
    for a, b in zip(feats, want_f):
        close(a, b, 2e-4)
    close(outs[0], want_o[0], 1e-4)
    names = {ops.conv2d_kernel_name(ops.conv2d_desc(2, m.in_channels, m.out_channels, 95, 1024, *m.kernel_size,
                                                    m.stride, m.padding))
             for m in d.modules() if isinstance(m, ad._SNConv) and m.out_channels >= 32}
    assert all(n.startswith(("conv_mfma", "conv_p2d")) for n in names), names   # MFMA kernels: patch tiles or the ring form


@pytest.mark.parametrize("cin,cout,kh,kw,sh,sw,ph,pw,h,w", [
    (32, 32, 3, 3, 1, 1, 1, 1, 17, 200), (32, 64, 3, 4, 1, 2, 1, 1, 17, 200), (64, 128, 4, 4, 2, 2, 1, 1, 18, 130),
    (64, 128, 4, 4, 2, 2, 1, 1, 19, 131), (128, 128, 3, 4, 1, 2, 1, 1, 9, 16), (256, 512, 4, 4, 2, 2, 1, 1, 6, 4),
    (2, 32, 7, 7, 1, 1, 3, 3, 21, 70), (512, 1, 1, 8, 1, 1, 0, 3, 5, 16), (16, 48, 3, 3, 2, 2, 1, 1, 11, 23),
    (32, 16, 5, 3, 1, 1, 2, 1, 8, 40), (4, 8, 3, 4, 1, 2, 1, 1, 9, 20), (8, 16, 4, 4, 2, 2, 1, 1, 10, 12),
    (3, 5, 2, 2, 1, 1, 3, 2, 6, 7)])
def test_conv2d_backward_data(cin, cout, kh, kw, sh, sw, ph, pw, h, w):
    """dx of every Conv2d shape of the STFT discriminators (+ odd sizes) against autograd, with the fused
    LeakyReLU-gradient mask."""
    torch.manual_seed(cin + cout + kh)
    x = torch.randn(2, cin, h, w, requires_grad=True)
    wt = torch.randn(cout, cin, kh, kw) / (cin * kh * kw) ** 0.5
    pre = torch.randn(2, cin, h, w)                                # stands for the activation that produced x
    xin = F.leaky_relu(pre, 0.2).detach().requires_grad_(True)
    y = F.conv2d(xin, wt, None, stride=(sh, sw), padding=(ph, pw))
    dy = torch.randn_like(y)
    y.backward(dy)
    want_plain = xin.grad
    want_masked = want_plain * torch.where(xin.detach() > 0, 1.0, 0.2)
    d = ops.conv2d_desc(2, cin, cout, h, w, kh, kw, (sh, sw), (ph, pw))
    pk = ops.conv2d_pack_bwd(d, wt.to(DEV))
    close(ops.conv2d_bwd_data(d, dy.to(DEV), pk), want_plain, 2e-5)
    close(ops.conv2d_bwd_data(d, dy.to(DEV), pk, xin.detach().to(DEV), 0.2), want_masked, 2e-5)
    extra = torch.randn(2, cin, h, w)
    close(ops.conv2d_bwd_data(d, dy.to(DEV), pk, xin.detach().to(DEV), 0.2, add=extra.to(DEV)),
          (want_plain + extra) * torch.where(xin.detach() > 0, 1.0, 0.2), 2e-5)
    from audio_generation_amd._lib import IMPL_MFMA_BF16X3
    d3 = ops.conv2d_desc(2, cin, cout, h, w, kh, kw, (sh, sw), (ph, pw), 0, 0.2, IMPL_MFMA_BF16X3)
    close(ops.conv2d_bwd_data(d3, dy.to(DEV), ops.conv2d_pack_bwd(d3, wt.to(DEV))), want_plain, 2e-5)


@pytest.mark.parametrize("cin,cout,kh,kw,sh,sw,ph,pw,h,w", [
    (32, 32, 3, 3, 1, 1, 1, 1, 17, 200), (32, 64, 3, 4, 1, 2, 1, 1, 17, 200), (64, 128, 4, 4, 2, 2, 1, 1, 19, 131),
    (128, 128, 3, 4, 1, 2, 1, 1, 9, 16), (256, 512, 4, 4, 2, 2, 1, 1, 6, 4), (2, 32, 7, 7, 1, 1, 3, 3, 21, 70),
    (512, 1, 1, 8, 1, 1, 0, 3, 5, 16), (5, 7, 3, 2, 2, 1, 0, 1, 10, 9),
    # stride-1 "same" layers with whole 32-column items: the barrier-free LDS-DMA kernel, one shape per tile config,
    # one- and two-row images (every item on the element-by-element edge path or next to it)
    (32, 32, 3, 3, 1, 1, 1, 1, 9, 64), (16, 64, 3, 3, 1, 1, 1, 1, 5, 32), (2, 32, 7, 7, 1, 1, 3, 3, 12, 96),
    (128, 128, 3, 3, 1, 1, 1, 1, 6, 32), (3, 40, 3, 3, 1, 1, 1, 1, 2, 64), (160, 130, 3, 3, 1, 1, 1, 1, 1, 32),
    (2, 20, 5, 5, 1, 1, 2, 2, 7, 128), (24, 200, 1, 1, 1, 1, 0, 0, 4, 64),
    # column-strided layers with whole 32-column items: the same kernel over the column-phase planes of x
    (32, 64, 3, 4, 1, 2, 1, 1, 17, 128), (64, 128, 4, 4, 2, 2, 1, 1, 10, 64), (16, 32, 3, 4, 1, 2, 1, 1, 5, 64),
    (40, 140, 4, 4, 2, 2, 1, 1, 2, 128), (8, 16, 3, 4, 1, 2, 1, 1, 1, 64),
    # long runs of interior items on the shared-operand kernel (two LDS slots, one barrier per item)
    (128, 256, 3, 3, 1, 1, 1, 1, 12, 128), (96, 128, 3, 4, 1, 2, 1, 1, 9, 256), (64, 128, 4, 4, 2, 2, 1, 1, 14, 128)])
@pytest.mark.parametrize("spectral", [False, True])
def test_conv2d_backward_weight(cin, cout, kh, kw, sh, sw, ph, pw, h, w, spectral):
    """dW / dbias against autograd; with spectral norm the gradient w.r.t. weight_orig (sigma = u.Wv, u / v fixed)."""
    torch.manual_seed(cin + cout + kw)
    x = torch.randn(3, cin, h, w)
    wt = (torch.randn(cout, cin, kh, kw) / (cin * kh * kw) ** 0.5).requires_grad_(True)
    b = torch.randn(cout, requires_grad=True)
    u, v = F.normalize(torch.randn(cout), dim=0), F.normalize(torch.randn(cin * kh * kw), dim=0)
    if spectral:
        sigma = torch.dot(u, torch.mv(wt.reshape(cout, -1), v))
        y = F.conv2d(x, wt / sigma, b, stride=(sh, sw), padding=(ph, pw))
    else:
        y = F.conv2d(x, wt, b, stride=(sh, sw), padding=(ph, pw))
    dy = torch.randn_like(y)
    y.backward(dy)
    d = ops.conv2d_desc(3, cin, cout, h, w, kh, kw, (sh, sw), (ph, pw))
    if spectral:
        sg = sigma.detach().reshape(1).to(DEV)
        dw, db = ops.conv2d_bwd_weight(d, x.to(DEV), dy.to(DEV), wt.detach().to(DEV), sg, u.to(DEV), v.to(DEV))
    else:
        dw, db = ops.conv2d_bwd_weight(d, x.to(DEV), dy.to(DEV))
    close(dw, wt.grad, 1e-4)
    close(db, b.grad, 2e-5)
    if not spectral:   # the bf16x3 contraction
        from audio_generation_amd._lib import IMPL_MFMA_BF16X3
        d3 = ops.conv2d_desc(3, cin, cout, h, w, kh, kw, (sh, sw), (ph, pw), 0, 0.2, IMPL_MFMA_BF16X3)
        dw3, db3 = ops.conv2d_bwd_weight(d3, x.to(DEV), dy.to(DEV))
        close(dw3, wt.grad, 1e-4)
        close(db3, b.grad, 2e-5)


@pytest.mark.parametrize("win,length", [(64, 500), (256, 1024), (1024, 24000)])
def test_stft_backward_is_the_adjoint(win, length):
    torch.manual_seed(win)
    x = torch.randn(2, length, requires_grad=True)
    y = od.stft_two_sided(x, win, win // 4)
    dy = torch.randn_like(y)
    y.backward(dy)
    close(ops.stft_backward(dy.to(DEV), length, win, True), x.grad, 2e-5)


def test_small_backward_kernels():
    torch.manual_seed(1)
    for scale, length in ((1, 100), (2, 101), (4, 1000)):
        x = torch.randn(3, 2, length, requires_grad=True)
        y = F.avg_pool1d(x, 2 * scale, stride=scale, padding=scale)
        dy, extra = torch.randn_like(y), torch.randn(3, 2, length)
        y.backward(dy)
        close(ops.avgpool1d_backward(dy.to(DEV), length, 2 * scale, scale, scale), x.grad, 1e-6)
        close(ops.avgpool1d_backward(dy.to(DEV), length, 2 * scale, scale, scale, add=extra.to(DEV)), x.grad + extra, 1e-6)
    z = torch.randn(4, 1, 77, requires_grad=True)
    s = torch.sigmoid(z)
    dy = torch.randn_like(s)
    s.backward(dy)
    close(ops.sigmoid_backward(dy.to(DEV), s.detach().to(DEV)), z.grad, 1e-6)
    w = torch.randn(24, 5, 7, requires_grad=True)
    u, v = F.normalize(torch.randn(24), dim=0), F.normalize(torch.randn(35), dim=0)
    sigma = torch.dot(u, torch.mv(w.reshape(24, -1), v))
    gn = torch.randn(24, 5, 7)
    (w / sigma).backward(gn)
    got = ops.spectral_grad_(gn.clone().to(DEV), w.detach().to(DEV), sigma.detach().reshape(1).to(DEV), u.to(DEV), v.to(DEV))
    close(got, w.grad, 1e-5)


def test_stft_discriminator_backward_runs_on_the_native_kernels(monkeypatch):
    """No ATen bridge on the STFT discriminator: 14 layers x 3 passes of dW calls, bwd-data everywhere but the
    first conv of the two passes whose input needs no gradient (real, detached fake)."""
    torch.manual_seed(5)
    d = ad.STFTDiscriminator(first_channel_size=16, win_length=256).to(DEV).train()
    calls = {"dw": 0, "dx": 0, "bridge": 0}
    real_w, real_x, real_f = ops.conv2d_bwd_weight, ops.conv2d_bwd_data, ops.conv2d_bwd_data_fewchannels
    monkeypatch.setattr(ops, "conv2d_bwd_data_fewchannels",
                        lambda *a, **k: (calls.__setitem__("dx", calls["dx"] + 1), real_f(*a, **k))[1])
    monkeypatch.setattr(ops, "conv2d_bwd_weight", lambda *a, **k: (calls.__setitem__("dw", calls["dw"] + 1), real_w(*a, **k))[1])
    monkeypatch.setattr(ops, "conv2d_bwd_data", lambda *a, **k: (calls.__setitem__("dx", calls["dx"] + 1), real_x(*a, **k))[1])
    assert not hasattr(ad, "_MultiOutBridge")      # there is no ATen bridge any more
    orig = 0.3 * torch.randn(2, 1, 4096, device=DEV)
    rec = (orig + 0.05 * torch.randn_like(orig)).requires_grad_(True)
    gl, dl = ad.discriminator_generator_loss(orig, rec, d)
    (gl + dl).backward()
    assert calls["bridge"] == 0 and calls["dw"] == 3 * 14 and calls["dx"] == 14 + 2 * 13
    assert rec.grad is not None and all(p.grad is not None for p in d.parameters())


@pytest.mark.parametrize("cin,cout,k,s,g,pad,length", [(16, 64, 41, 4, 4, 0, 3000), (64, 256, 41, 4, 16, 0, 900),
                                                         (512, 1024, 41, 4, 256, 0, 200), (6, 9, 4, 3, 3, 2, 50),
                                                         (8, 8, 5, 1, 2, 1, 33)])
def test_grouped_conv1d_backward(cin, cout, k, s, g, pad, length):
    torch.manual_seed(cin + k)
    pre = torch.randn(2, cin, length)
    x = F.leaky_relu(pre, 0.2).detach().requires_grad_(True)
    w = (torch.randn(cout, cin // g, k) / (cin // g * k) ** 0.5).requires_grad_(True)
    b = torch.randn(cout, requires_grad=True)
    sigma = torch.tensor([0.37])
    y = F.conv1d(x, w / sigma, b, stride=s, padding=pad, groups=g)
    dz = torch.randn_like(y)
    y.backward(dz)
    d = ops.conv_desc(CONV_PADDED, 2, cin, cout, length, k, s, 1, 0, 0.2, 0, groups=g, padding=pad)
    extra = torch.randn(2, cin, length)
    dx = ops.conv_grouped_bwd_data(d, dz.to(DEV), w.detach().to(DEV), sigma.to(DEV), extra.to(DEV), x.detach().to(DEV), 0.2)
    close(dx, (x.grad + extra) * torch.where(x.detach() > 0, 1.0, 0.2), 2e-5)
    dw, db = ops.conv_grouped_bwd_weight(d, x.detach().to(DEV), dz.to(DEV))
    close(dw, w.grad * sigma, 1e-4)          # plain gradient (w.r.t. w / sigma)
    close(db, b.grad, 2e-5)


def test_waveform_discriminator_backward_runs_on_the_native_kernels(monkeypatch):
    torch.manual_seed(6)
    d = ad.WaveFormDiscriminator(1, n_blocks=2).to(DEV).train()
    calls = {"bridge": 0, "grouped": 0}
    assert not hasattr(ad, "_MultiOutBridge")      # there is no ATen bridge any more
    real_g = ops.conv_grouped_bwd_weight
    monkeypatch.setattr(ops, "conv_grouped_bwd_weight", lambda *a, **k: (calls.__setitem__("grouped", calls["grouped"] + 1), real_g(*a, **k))[1])
    orig = 0.3 * torch.randn(2, 1, 16384, device=DEV)
    rec = (orig + 0.05 * torch.randn_like(orig)).requires_grad_(True)
    gl, dl = ad.discriminator_generator_loss(orig, rec, d)
    (gl + dl).backward()
    assert calls["bridge"] == 0 and calls["grouped"] == 3 * 2 * 4
    assert rec.grad is not None and all(p.grad is not None for p in d.parameters())


def test_bf16x3_stft_discriminator_matches_fp32():
    """STFT discriminator with its Conv2d layers on the bf16x3 kernels: outputs, features, loss and gradients agree
    with the fp32 run to fp32-class tolerances (full widths, win 512)."""
    torch.manual_seed(9)
    d = ad.STFTDiscriminator(win_length=512).to(DEV).train()
    orig = 0.3 * torch.randn(2, 1, 12000, device=DEV)
    with torch.no_grad():          # let the power iteration converge: a fresh sigma = u.Wv is tiny, the activations
        for _ in range(12):        # huge, the sigmoid saturated and the gradients ill-conditioned
            d(orig)
    sd = {k: v.clone() for k, v in d.state_dict().items()}
    rec0 = orig + 0.05 * torch.randn_like(orig)
    res = {}
    for mode in ("fp32", "bf16x3"):
        d.load_state_dict(sd)
        ad.set_arithmetic(d, mode)
        rec = rec0.clone().requires_grad_(True)
        for p_ in d.parameters():
            p_.grad = None
        gl, dl = ad.discriminator_generator_loss(orig, rec, d)
        (gl + dl).backward()
        res[mode] = (float(gl), float(dl), rec.grad.clone(), [p_.grad.clone() for p_ in d.parameters()])
    ad.set_arithmetic(d, "fp32")
    a, b = res["fp32"], res["bf16x3"]
    assert abs(a[0] - b[0]) <= 1e-4 * abs(a[0]) and abs(a[1] - b[1]) <= 1e-5 * abs(a[1])
    # The kernels themselves agree with autograd to ~1e-6 in both arithmetics (test_conv2d / test_conv2d_backward_data
    # run both); the gradient of a GAN discriminator through 14 layers, LeakyReLU masks and the hinge is
    # ill-conditioned, so whole-model gradients are compared by direction (cosine) and a loose max-norm.
    for ga, gb in [(a[2], b[2])] + list(zip(a[3], b[3])):
        cos = float((ga * gb).sum() / (ga.norm() * gb.norm() + 1e-30))
        assert cos > 0.9999, cos
        close(gb, ga, 5e-2)
    names = {ops.conv2d_kernel_name(m.desc2d(torch.empty(2, m.in_channels, 20, 64))) for m in d.modules()
             if isinstance(m, ad._SNConv)}
    assert names


@pytest.mark.parametrize("cin,cout,kh,kw,ph,pw,h,w", [(2, 32, 7, 7, 3, 3, 21, 70), (3, 48, 5, 5, 2, 2, 9, 33), (2, 32, 7, 7, 3, 3, 40, 1024),
                                                      (1, 16, 3, 9, 1, 4, 6, 20)])
def test_conv2d_backward_data_few_input_channels(cin, cout, kh, kw, ph, pw, h, w):
    """The column-split backward-data of the first (2-channel) conv against autograd, with the fused add."""
    torch.manual_seed(cin + cout)
    x = torch.randn(2, cin, h, w, requires_grad=True)
    wt = torch.randn(cout, cin, kh, kw) / (cin * kh * kw) ** 0.5
    sigma = torch.tensor([1.7])
    y = F.conv2d(x, wt / sigma, None, padding=(ph, pw))
    dy, extra = torch.randn_like(y), torch.randn(2, cin, h, w)
    y.backward(dy)
    d = ops.conv2d_desc(2, cin, cout, h, w, kh, kw, (1, 1), (ph, pw))
    got = ops.conv2d_bwd_data_fewchannels(d, dy.to(DEV), wt.to(DEV), sigma.to(DEV), extra.to(DEV))
    close(got, x.grad + extra, 2e-5)


def test_weight_normalised_discriminators_match_torch():
    """norm="weight" (add_util_norm's other branch, utils.py:34-42; the default is spectral): forward, loss and every
    gradient (weight_g / weight_v / bias, and the input) against the same stacks built from torch modules with
    torch.nn.utils.weight_norm on the CPU."""
    import warnings
    torch.manual_seed(3)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        wave = ad.WaveFormDiscriminator(1, n_blocks=2, norm="weight").to(DEV).train()
        stft = ad.STFTDiscriminator(first_channel_size=8, win_length=128, norm="weight").to(DEV).train()
    assert sorted(k for k in wave.state_dict() if "layers.1" in k and k.startswith("layers.0."))[:3] == \
        ["layers.0.layers.1.0.bias", "layers.0.layers.1.0.weight_g", "layers.0.layers.1.0.weight_v"]
    x = (0.3 * torch.randn(2, 1, 32768)).to(DEV)
    y = (x + 0.05 * torch.randn_like(x)).requires_grad_(True)
    for disc in (wave, stft):
        gl, dl = ad.discriminator_generator_loss(x, y, disc)
        (gl + dl).backward()
        got = {n: p.grad.detach().cpu().clone() for n, p in disc.named_parameters()}
        gy = y.grad.detach().cpu().clone()
        for p in disc.parameters():
            p.grad = None
        y.grad = None
        # reference: the ATen restatement of the same module, differentiated by autograd on the CPU
        sd = {k: v.detach().cpu() for k, v in disc.state_dict().items()}
        params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k.endswith(("weight_g", "weight_v", "bias"))}

        def wn(prefix):
            return torch._weight_norm(params[prefix + "weight_v"], params[prefix + "weight_g"], 0)

        def ref_wave(inp):
            outs, feats = [], []
            for b in range(2):
                h = F.avg_pool1d(inp, 2 * 2 ** b, stride=2 ** b, padding=2 ** b)
                feats.append(h)
                for i in range(7):
                    last = i == 6
                    pre = f"layers.{b}.layers.{i + 1}." + ("" if last else "0.")
                    h = F.conv1d(h, wn(pre), params[pre + "bias"], stride=od.WAVE_STRIDES[i], groups=od.WAVE_GROUPS[i])
                    if not last:
                        h = F.leaky_relu(h, 0.2)
                    feats.append(h)
                outs.append(torch.sigmoid(h))
            return outs, feats

        def ref_stft(inp):
            h = od.stft_two_sided(inp.squeeze(1), 128, 32, True)
            h = F.conv2d(h, wn("first_conv."), params["first_conv.bias"], padding=3)
            feats = [h]
            for i, stride in enumerate(od.STFT_STRIDES):
                pre = f"blocks.{i}.layers."
                h = F.leaky_relu(F.conv2d(h, wn(pre + "0."), params[pre + "0.bias"], padding=1), 0.2)
                k = (stride[0] + 2, stride[1] + 2)
                h = F.conv2d(h, wn(pre + "2."), params[pre + "2.bias"], stride=stride, padding=((k[0] - 1) // 2, (k[1] - 1) // 2))
                feats.append(h)
            h = F.conv2d(h, wn("final_conv."), params["final_conv.bias"], padding=(0, (128 // 128 - 1) // 2))
            return [torch.sigmoid(h)], feats

        xc, yc = x.cpu(), y.detach().cpu().requires_grad_(True)
        rgl, rdl = od.discriminator_generator_loss(xc, yc, ref_wave if disc is wave else ref_stft)
        (rgl + rdl).backward()
        assert abs(float(gl) - float(rgl)) <= 2e-4 * abs(float(rgl)) and abs(float(dl) - float(rdl)) <= 2e-4 * abs(float(rdl))
        # (the L1 feature loss makes these gradients ill-conditioned -- sign flips of near-zero differences -- so: direction
        # to 1e-4, element-wise to 5 % of the largest entry (as in the bf16x3 comparison above); the per-kernel tests above carry the tight tolerances)
        def same(a, b):
            a, b = a.double().flatten().cpu(), b.double().flatten()
            assert float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-300)) > 0.9999
            close(a.float(), b.float(), 5e-2)
        same(gy, yc.grad)
        for n, g in got.items():
            same(g, params[n].grad)


@pytest.mark.parametrize("n", [1, 7, 4096, 1 << 20, (1 << 22) + 1027, 3 * 5 * 7 * 11 * 13 * 17 * 4 + 3])
@pytest.mark.parametrize("mode", [ops.REDUCE_MEAN, ops.REDUCE_HINGE_REAL, ops.REDUCE_HINGE_FAKE, ops.REDUCE_L1,
                                  ops.REDUCE_ABS_EPS])
def test_reductions_16_byte_path_edges_and_misaligned_views(n, mode):
    """The 16-byte-load form of the loss reductions: lengths that are not multiples of 4 (scalar tail), lengths below one
    vector, and operands that start 4 bytes into an allocation (the 4-byte path) -- against float64 of the same term."""
    g = torch.Generator().manual_seed(n % 1000 + mode)
    for off in (0, 1):
        xb = torch.randn(n + 1, generator=g).to(DEV)
        yb = torch.randn(n + 1, generator=g).to(DEV)
        x, y = xb[off:off + n], yb[off:off + n]
        two = mode == ops.REDUCE_L1
        out = ops.reduce_mean(x, mode, y if two else None)
        xd, yd = x.double(), y.double()
        ref = {ops.REDUCE_MEAN: xd, ops.REDUCE_HINGE_REAL: torch.clamp(xd - 1, max=0),
               ops.REDUCE_HINGE_FAKE: torch.clamp(-xd - 1, max=0), ops.REDUCE_L1: (xd - yd).abs(),
               ops.REDUCE_ABS_EPS: (xd.float() + 1e-3).double().abs()}[mode].mean()
        assert abs(float(out) - float(ref)) <= 2e-6 * max(1.0, abs(float(ref))) + 2e-7
        gr = torch.tensor([0.75], device=DEV)
        dx, dy = ops.reduce_mean_backward(x, mode, gr, y if two else None, two)
        s = 0.75 * float(torch.tensor(1.0 / n, dtype=torch.float64).float())
        want = {ops.REDUCE_MEAN: torch.ones_like(x), ops.REDUCE_HINGE_REAL: (x - 1 < 0).float(),
                ops.REDUCE_HINGE_FAKE: -((-x - 1) < 0).float(), ops.REDUCE_L1: torch.sign(x - y),
                ops.REDUCE_ABS_EPS: torch.sign(x + 1e-3)}[mode] * torch.tensor(0.75, device=DEV) * float(
                    torch.tensor(1.0 / n, dtype=torch.float64).float())
        assert torch.allclose(dx, want, rtol=1e-6, atol=0), (n, mode, off, s)
        if two:
            assert torch.equal(dy, -dx)


@pytest.mark.parametrize("shape", [(3, 5, 7, 9), (2, 32, 70, 64), (1, 16, 72000), (4, 1027)])
def test_feature_means_equal_the_two_separate_means_bit_for_bit(shape):
    """agx_feature_means: one pass over (x, y) for mean|x - y| and mean|x + 1e-3| -- values and gradients are those of
    agx_reduce_mean modes 3 and 4 (and the sum of their two gradients), bit for bit (discriminator.py:236-243)."""
    torch.manual_seed(sum(shape))
    x, y = torch.randn(*shape, device=DEV), torch.randn(*shape, device=DEV)
    y.view(-1)[::5] = x.view(-1)[::5]                        # exact ties: sign(0) = 0
    x.view(-1)[1::7] = -1e-3                                 # ... and |x + 1e-3| at its kink
    pair = ops.feature_means(x, y)
    assert float(pair[0]) == float(ops.reduce_mean(x, ops.REDUCE_L1, y))
    assert float(pair[1]) == float(ops.reduce_mean(x, ops.REDUCE_ABS_EPS))
    g = torch.tensor([0.37, -1.9], device=DEV)
    dx, dy = ops.feature_means_backward(x, y, g)
    dx1, dy1 = ops.reduce_mean_backward(x, ops.REDUCE_L1, g[0:1].clone(), y, True)
    dx2, _ = ops.reduce_mean_backward(x, ops.REDUCE_ABS_EPS, g[1:2].clone())
    assert torch.equal(dx, dx1 + dx2) and torch.equal(dy, dy1)
    only_dy = ops.feature_means_backward(x, y, g, want_dx=False)
    assert only_dy[0] is None and torch.equal(only_dy[1], dy1)


def test_feature_matching_term_through_autograd_matches_the_two_mean_form():
    """discriminator_generator_loss's scaled feature term built from _FeatureMeans vs the same term built from the two
    separate _mean calls: loss and both gradients identical."""
    from audio_generation_amd.discriminator import _FeatureMeans, _mean
    torch.manual_seed(5)
    x0, y0 = torch.randn(2, 8, 33, 17, device=DEV), torch.randn(2, 8, 33, 17, device=DEV)
    res = []
    for fused in (True, False):
        x, y = x0.clone().requires_grad_(True), y0.clone().requires_grad_(True)
        if fused:
            pair = _FeatureMeans.apply(x, y)
            loss = pair[0] / 13 / pair[1]
        else:
            loss = _mean(ops.REDUCE_L1, x, y) / 13 / _mean(ops.REDUCE_ABS_EPS, x)
        (100 * loss).backward()
        res.append((loss.detach(), x.grad, y.grad))
    assert float(res[0][0]) == float(res[1][0])
    assert torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])
