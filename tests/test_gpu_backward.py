"""Native backward kernels against the CPU oracle's autograd."""
import pytest
import torch

from audio_generation_amd import _lib, ops
from oracle import codec
from tests.helpers import max_abs
from tests.test_gpu_parity import KIND, SHAPES

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _oracle_conv(kind, x, w, bias, s, d):
    if kind == "conv":
        return codec.causal_conv1d(x, w, bias, stride=s, dilation=d)
    if kind == "convt":
        return codec.causal_conv_t1d(x, w, bias, stride=s)
    return codec.upsample_conv1d(x, w, bias, s)


@pytest.mark.parametrize("impl", ["direct", "mfma"])
def test_backward_data_matches_autograd(impl):
    gen = torch.Generator().manual_seed(31)
    checked = 0
    for (kind, cin, cout, k, s, d, b, length) in SHAPES:
        # the backward op has the channel roles swapped: MFMA tiles need fwd-Cout % 16 == 0 and >= 32 rows
        rows = cin * (s if kind == "conv" and s > 1 else 1)
        if impl == "mfma" and (cout % 16 != 0 or rows < 32):
            continue
        wshape = (cin, cout, k) if kind == "convt" else (cout, cin, k)
        v = torch.randn(wshape, generator=gen) / (cin * k) ** 0.5
        g = torch.rand((wshape[0], 1, 1), generator=gen) + 0.5
        x = torch.randn(b, cin, length, generator=gen, requires_grad=True)
        y = _oracle_conv(kind, x, codec.fold_weight_norm(g, v), None, s, d)
        dy = torch.randn(y.shape, generator=gen)
        (want,) = torch.autograd.grad(y, x, dy)
        desc = ops.conv_desc(KIND[kind], b, cin, cout, length, k, s, d, 0, 0.1,
                             _lib.IMPL_DIRECT if impl == "direct" else _lib.IMPL_MFMA)
        packed = ops.conv_pack_bwd(desc, v.to(DEV), g.to(DEV))
        dx = ops.conv_bwd_data(desc, dy.to(DEV), packed)
        assert tuple(dx.shape) == tuple(want.shape)
        err = max_abs(dx.cpu(), want)
        assert err < 3e-5 * max(1.0, float(want.abs().max())), (impl, kind, cin, cout, k, s, d, err)
        checked += 1
    assert checked >= (len(SHAPES) if impl == "direct" else 12)


def test_backward_data_residual_and_activation_gradient():
    """dx = leaky'(saved input) * (add + conv_bwd(dy)) -- the fused form the residual block needs."""
    gen = torch.Generator().manual_seed(32)
    b, c, length, k, d = 2, 32, 400, 7, 3
    v = torch.randn(c, c, k, generator=gen) / (c * k) ** 0.5
    g = torch.rand(c, 1, 1, generator=gen) + 0.5
    pre = torch.randn(b, c, length, generator=gen, requires_grad=True)   # previous layer's pre-activation
    x = codec.leaky(pre)                                                 # what the forward saved
    y = x + codec.causal_conv1d(x, codec.fold_weight_norm(g, v), None, dilation=d)
    dy = torch.randn(y.shape, generator=gen)
    (want,) = torch.autograd.grad(y, pre, dy)
    desc = ops.conv_desc(_lib.CONV_CAUSAL, b, c, c, length, k, 1, d)
    packed = ops.conv_pack_bwd(desc, v.to(DEV), g.to(DEV))
    dx = ops.conv_bwd_data(desc, dy.to(DEV), packed, add=dy.to(DEV), mask=x.detach().to(DEV), slope=0.1)
    assert max_abs(dx.cpu(), want) < 3e-5 * max(1.0, float(want.abs().max()))


# longer rows (the DMA path of the barrier-free kernel over many interior chunks of the phase-split copies), odd
# strides with a dilation, lengths that are no multiple of the stride, a single-chunk row
DW_EXTRA = [("conv", 32, 64, 5, 2, 1, 3, 5001), ("conv", 24, 40, 7, 3, 2, 2, 3001), ("conv", 64, 128, 9, 4, 1, 1, 4100),
            ("upconv", 64, 32, 5, 2, 1, 2, 2100), ("convt", 32, 16, 9, 4, 1, 2, 1500), ("conv", 16, 16, 3, 5, 1, 2, 31),
            ("conv", 160, 130, 3, 1, 2, 1, 2050), ("upconv", 40, 24, 7, 3, 1, 1, 999)]


def test_backward_weight_bias_and_weight_norm_match_autograd():
    gen = torch.Generator().manual_seed(33)
    checked = 0
    for (kind, cin, cout, k, s, d, b, length) in SHAPES + DW_EXTRA:
        wshape = (cin, cout, k) if kind == "convt" else (cout, cin, k)
        v = (torch.randn(wshape, generator=gen) / (cin * k) ** 0.5).requires_grad_(True)
        g = (torch.rand((wshape[0], 1, 1), generator=gen) + 0.5).requires_grad_(True)
        bias = (torch.randn(cout, generator=gen) * 0.1).requires_grad_(True)
        x = torch.randn(b, cin, length, generator=gen)
        y = _oracle_conv(kind, x, codec.fold_weight_norm(g, v), bias, s, d)
        dy = torch.randn(y.shape, generator=gen)
        want_v, want_g, want_b = torch.autograd.grad(y, (v, g, bias), dy)
        desc = ops.conv_desc(KIND[kind], b, cin, cout, length, k, s, d)
        dv, dg, db = ops.conv_bwd_weight(desc, x.to(DEV), dy.to(DEV), v.detach().to(DEV), g.detach().to(DEV))
        for got, want, nm in ((dv, want_v, "dv"), (dg, want_g, "dg"), (db, want_b, "db")):
            err, scale = max_abs(got.cpu(), want), float(want.abs().max())
            assert err < 2e-4 * max(1.0, scale), (kind, cin, cout, k, s, d, nm, err, scale)
        # plain (not weight-normed) weights: dv is the weight gradient
        w_plain = v.detach().clone().requires_grad_(True)
        y2 = _oracle_conv(kind, x, w_plain, None, s, d)
        (want_w,) = torch.autograd.grad(y2, w_plain, dy)
        dw, none_g, none_b = ops.conv_bwd_weight(desc, x.to(DEV), dy.to(DEV), w_plain.detach().to(DEV), None,
                                                 want_bias=False)
        assert none_g is None and none_b is None
        assert max_abs(dw.cpu(), want_w) < 2e-4 * max(1.0, float(want_w.abs().max())), (kind, cin, cout, k, s, d)
        # the bf16x3 contraction (descriptor impl = AGX_IMPL_MFMA_BF16X3)
        desc3 = ops.conv_desc(KIND[kind], b, cin, cout, length, k, s, d, impl=_lib.IMPL_MFMA_BF16X3)
        dw3, _, _ = ops.conv_bwd_weight(desc3, x.to(DEV), dy.to(DEV), w_plain.detach().to(DEV), None, want_bias=False)
        assert max_abs(dw3.cpu(), want_w) < 2e-4 * max(1.0, float(want_w.abs().max())), (kind, cin, cout, k, s, d)
        checked += 1
    assert checked == len(SHAPES) + len(DW_EXTRA)


@pytest.mark.parametrize("channelwise", [True, False])
@pytest.mark.parametrize("scale,n_points,length", [(2, 16, 37), (5, 40, 300), (4, 8, 1), (1, 4, 19)])
def test_wavelet_fold_backward_matches_autograd(channelwise, scale, n_points, length):
    """dh and d(wavelet_scale) of the fold (two-tap collapsed form) against autograd through the oracle's
    (B,C,L,n_points) expansion, incl. the raw-sample tail, L = 1 and the shared-sigma variant."""
    from oracle import wavelets as ow
    torch.manual_seed(scale * 10 + length)
    b, c = 3, 6
    h = torch.randn(b, c, length, requires_grad=True)
    space = torch.linspace(-10, 10, n_points)
    sigma = (40.0 + 5 * torch.rand(1, c, 1, 1) if channelwise else torch.tensor(33.0)).requires_grad_(True)
    dout = torch.randn(b, c, length * scale)
    out = ow.wavelet_fold(h, space, sigma, scale)
    out.backward(dout)
    dh, dsig = ops.wavelet_fold_backward(h.detach().to(DEV), dout.to(DEV), space.to(DEV), sigma.detach().to(DEV), scale)
    assert dsig.shape == sigma.shape
    assert max_abs(dh.cpu(), h.grad) <= 1e-5 * float(h.grad.abs().max())
    assert max_abs(dsig.cpu(), sigma.grad) <= 1e-4 * float(sigma.grad.abs().max()) + 1e-9
