"""Conv2d layers of the STFT discriminators on the persistent ring kernel (csrc/conv_p.hip, D2 geometries): forward
of the kh x 3 stride-1 and kh x 4 column-stride-2 layers, backward-data of the stride-1 layers (flipped kernel, with the
fused gradient add and LeakyReLU-gradient mask) -- against torch's conv2d / autograd on the CPU, and against the
patch-tile kernel (knob conv_impl = 0) they replace.  discriminator.py:87-197 of the reference."""
import pytest
import torch
import torch.nn.functional as F

from audio_generation_amd import _lib, ops
from audio_generation_amd._lib import EPI_LEAKY_PRE
from tests.helpers import max_abs

pytestmark = pytest.mark.gpu
DEV = "cuda"


def close(got, want, tol):
    scale = max(1.0, float(want.abs().max()))
    assert max_abs(got.detach().cpu(), want) <= tol * scale, max_abs(got.detach().cpu(), want) / scale


@pytest.fixture
def knob():
    lib = _lib.load()
    yield lambda v: lib.agx_set_tuning(b"conv_impl", v)
    lib.agx_set_tuning(b"conv_impl", 1)


FWD = [  # cin, cout, kh, kw, sh, sw, h, w, variant
    (64, 64, 3, 3, 1, 1, 7, 250, "conv_p2d<k3,64x256>"), (64, 64, 3, 3, 1, 1, 3, 512, "conv_p2d<k3,64x256>"),
    (128, 128, 3, 3, 1, 1, 6, 250, "conv_p2d<k3,128x128>"), (128, 128, 3, 3, 1, 1, 5, 253, "conv_p2d<k3,128x128>"),
    (256, 256, 3, 3, 1, 1, 5, 120, "conv_p2d<k3,128x128>"), (32, 128, 3, 3, 1, 1, 1, 128, "conv_p2d<k3,128x128>"),
    (32, 64, 3, 4, 1, 2, 6, 500, "conv_p2d<k4s2,64x256>"), (64, 128, 4, 4, 2, 2, 10, 256, "conv_p2d<k4s2,128x128>"),
    (64, 128, 4, 4, 2, 2, 11, 250, "conv_p2d<k4s2,128x128>"), (128, 128, 3, 4, 1, 2, 5, 480, "conv_p2d<k4s2,128x128>"),
    (256, 512, 4, 4, 2, 2, 6, 241, "conv_p2d<k4s2,128x128>"), (64, 128, 5, 3, 2, 1, 9, 125, "conv_p2d<k3,128x128>"),
    # narrow maps: several output rows per tile (columns per tile row = 64 / 32 / 128 ...)
    (128, 128, 3, 3, 1, 1, 9, 64, "conv_p2d<k3,128x128>"), (256, 256, 3, 3, 1, 1, 7, 32, "conv_p2d<k3,128x128>"),
    (128, 128, 3, 3, 1, 1, 6, 50, "conv_p2d<k3,128x128>"), (64, 64, 3, 3, 1, 1, 9, 128, "conv_p2d<k3,64x256>"),
    (64, 64, 3, 3, 1, 1, 5, 64, "conv_p2d<k3,64x256>"), (32, 32, 3, 3, 1, 1, 9, 128, "conv_p2d<k3,32x512>"),
    (128, 256, 4, 4, 2, 2, 10, 128, "conv_p2d<k4s2,128x128>"), (256, 256, 3, 4, 1, 2, 7, 64, "conv_p2d<k4s2,128x128>"),
    (32, 64, 3, 4, 1, 2, 11, 128, "conv_p2d<k4s2,64x256>"),
]


@pytest.mark.parametrize("cin,cout,kh,kw,sh,sw,h,w,variant", FWD)
def test_forward_on_the_ring_kernel(cin, cout, kh, kw, sh, sw, h, w, variant, knob):
    torch.manual_seed(cin + cout + h + w)
    ph, pw = (kh - 1) // 2, 1
    x = torch.randn(3, cin, h, w)
    wt = torch.randn(cout, cin, kh, kw) / (cin * kh * kw) ** 0.5
    b = torch.randn(cout)
    sigma = torch.tensor([1.7])
    want = F.leaky_relu(F.conv2d(x, wt / 1.7, b, stride=(sh, sw), padding=(ph, pw)), 0.2)
    d = ops.conv2d_desc(3, cin, cout, h, w, kh, kw, (sh, sw), (ph, pw), EPI_LEAKY_PRE, 0.2)
    assert ops.conv2d_kernel_name(d) == variant
    got = ops.conv2d_forward(d, x.to(DEV), ops.conv2d_pack(d, wt.to(DEV), sigma.to(DEV)), b.to(DEV))
    close(got, want, 1e-5)
    knob(0)
    assert ops.conv2d_kernel_name(d).startswith("conv_mfma")
    old = ops.conv2d_forward(d, x.to(DEV), ops.conv2d_pack(d, wt.to(DEV), sigma.to(DEV)), b.to(DEV))
    close(old, want, 1e-5)
    # no bias, no activation
    knob(1)
    d0 = ops.conv2d_desc(3, cin, cout, h, w, kh, kw, (sh, sw), (ph, pw))
    got0 = ops.conv2d_forward(d0, x.to(DEV), ops.conv2d_pack(d0, wt.to(DEV)), None)
    close(got0, F.conv2d(x, wt, None, stride=(sh, sw), padding=(ph, pw)), 1e-5)


@pytest.mark.parametrize("cin,cout,kh,h,w,variant", [
    (64, 64, 3, 7, 250, "conv_p2d<k3,64x256>"), (128, 128, 3, 6, 250, "conv_p2d<k3,128x128>"),
    (128, 128, 3, 5, 253, "conv_p2d<k3,128x128>"), (256, 256, 3, 5, 120, "conv_p2d<k3,128x128>"),
    (128, 64, 3, 4, 500, "conv_p2d<k3,128x128>"), (128, 64, 5, 9, 125, "conv_p2d<k3,128x128>"),
    (128, 128, 3, 9, 64, "conv_p2d<k3,128x128>"), (256, 256, 3, 7, 32, "conv_p2d<k3,128x128>"),
    (128, 128, 3, 6, 50, "conv_p2d<k3,128x128>"), (64, 64, 3, 5, 64, "conv_p2d<k3,64x256>")])
def test_backward_data_on_the_ring_kernel(cin, cout, kh, h, w, variant, knob):
    """stride-1 layers: dx = conv(dy, flipped kernel); (cin, cout) are the FORWARD layer's."""
    torch.manual_seed(cin + cout + h + w)
    ph = (kh - 1) // 2
    pre = torch.randn(2, cin, h, w)
    xin = F.leaky_relu(pre, 0.2).detach().requires_grad_(True)
    wt = torch.randn(cout, cin, kh, 3) / (cin * kh * 3) ** 0.5
    y = F.conv2d(xin, wt / 0.8, None, padding=(ph, 1))
    dy = torch.randn_like(y)
    y.backward(dy)
    extra = torch.randn(2, cin, h, w)
    slope_mask = torch.where(xin.detach() > 0, 1.0, 0.2)
    d = ops.conv2d_desc(2, cin, cout, h, w, kh, 3, (1, 1), (ph, 1))
    assert ops.conv2d_bwd_data_kernel_name(d) == variant
    sigma = torch.tensor([0.8]).to(DEV)
    pk = ops.conv2d_pack_bwd(d, wt.to(DEV), sigma)
    close(ops.conv2d_bwd_data(d, dy.to(DEV), pk), xin.grad, 2e-5)
    close(ops.conv2d_bwd_data(d, dy.to(DEV), pk, xin.detach().to(DEV), 0.2), xin.grad * slope_mask, 2e-5)
    got = ops.conv2d_bwd_data(d, dy.to(DEV), pk, xin.detach().to(DEV), 0.2, add=extra.to(DEV))
    close(got, (xin.grad + extra) * slope_mask, 2e-5)
    knob(0)
    assert ops.conv2d_bwd_data_kernel_name(d).startswith("conv_mfma")
    old = ops.conv2d_bwd_data(d, dy.to(DEV), ops.conv2d_pack_bwd(d, wt.to(DEV), sigma), xin.detach().to(DEV), 0.2,
                              add=extra.to(DEV))
    close(old, (xin.grad + extra) * slope_mask, 2e-5)


def test_half_empty_column_blocks_stay_on_the_patch_tiles():
    """the ring takes a layer when its column blocks (whole rows of >= 32 columns for narrow maps) are >= 70 % full"""
    for (cin, cout, kh, kw, sh, sw, h, w, ring) in [(128, 128, 3, 3, 1, 1, 9, 16, False), (64, 64, 3, 3, 1, 1, 9, 20, False),
                                                   (32, 32, 3, 3, 1, 1, 9, 300, False), (64, 128, 4, 4, 2, 2, 18, 130, False),
                                                   (128, 128, 3, 3, 1, 1, 9, 1024 + 40, True), (128, 128, 3, 3, 1, 1, 9, 24, True)]:
        d = ops.conv2d_desc(2, cin, cout, h, w, kh, kw, (sh, sw), ((kh - 1) // 2, 1))
        name = ops.conv2d_kernel_name(d)
        assert name.startswith("conv_p2d") == ring, (name, cin, w)


def test_packed_image_does_not_depend_on_the_feature_map_size():
    """discriminator.py packs a layer once with a nominal (1, 64, 64) descriptor and applies the image to maps of any
    size: the image (tile image included) must be the same whatever size the descriptor names."""
    torch.manual_seed(0)
    for (cin, cout, kh, kw, sh, sw) in [(64, 64, 3, 3, 1, 1), (64, 128, 4, 4, 2, 2), (128, 128, 3, 4, 1, 2)]:
        wt = (torch.randn(cout, cin, kh, kw) / (cin * kh * kw) ** 0.5).to(DEV)
        nominal = ops.conv2d_desc(1, cin, cout, 64, 64, kh, kw, (sh, sw), (1, 1))
        real = ops.conv2d_desc(2, cin, cout, 6, 512, kh, kw, (sh, sw), (1, 1))
        pk_n, pk_r = ops.conv2d_pack(nominal, wt), ops.conv2d_pack(real, wt)
        assert pk_n.shape == pk_r.shape and torch.equal(pk_n, pk_r)
        assert ops.conv2d_kernel_name(real).startswith("conv_p2d")
        tiny = ops.conv2d_desc(1, cin, cout, 9, 8, kh, kw, (sh, sw), (1, 1))       # too narrow for the ring: same image still
        assert ops.conv2d_kernel_name(tiny).startswith("conv_mfma") and torch.equal(ops.conv2d_pack(tiny, wt), pk_r)
        x = torch.randn(2, cin, 6, 512)
        close(ops.conv2d_forward(real, x.to(DEV), pk_n, None), F.conv2d(x, wt.cpu(), None, stride=(sh, sw), padding=(1, 1)), 1e-5)
        if (sh, sw) == (1, 1):
            assert torch.equal(ops.conv2d_pack_bwd(nominal, wt), ops.conv2d_pack_bwd(real, wt))


@pytest.mark.parametrize("cin,cout,kh,sh,h,w,variant", [
    (32, 64, 3, 1, 6, 500, "conv_p2d<bwd s(1,2),64x256>"), (128, 128, 3, 1, 5, 480, "conv_p2d<bwd s(1,2),128x128>"),
    (64, 128, 4, 2, 10, 256, "conv_p2d<bwd s(2,2),128x128>"), (64, 128, 4, 2, 11, 250, "conv_p2d<bwd s(2,2),128x128>"),
    (256, 512, 4, 2, 6, 241, "conv_p2d<bwd s(2,2),128x128>"), (64, 32, 4, 2, 3, 256, "conv_p2d<bwd s(2,2),128x128>"),
    (128, 256, 4, 2, 10, 128, "conv_p2d<bwd s(2,2),128x128>"), (256, 256, 3, 1, 7, 64, "conv_p2d<bwd s(1,2),128x128>"),
    (32, 64, 3, 1, 11, 128, "conv_p2d<bwd s(1,2),64x256>"), (64, 128, 4, 2, 9, 60, "conv_p2d<bwd s(2,2),128x128>")])
def test_strided_backward_data_on_the_ring_kernel(cin, cout, kh, sh, h, w, variant, knob):
    """column stride 2 (kernels (3,4) stride (1,2) and (4,4) stride (2,2), padding 1): the phase GEMM on the ring for base
    positions 1 .. W/2 plus conv2d_bwd_first_cols_kernel for output column 0."""
    torch.manual_seed(cin + cout + h + w)
    pre = torch.randn(2, cin, h, w)
    xin = F.leaky_relu(pre, 0.2).detach().requires_grad_(True)
    wt = torch.randn(cout, cin, kh, 4) / (cin * kh * 4) ** 0.5
    y = F.conv2d(xin, wt / 1.3, None, stride=(sh, 2), padding=(1, 1))
    dy = torch.randn_like(y)
    y.backward(dy)
    extra = torch.randn(2, cin, h, w)
    slope_mask = torch.where(xin.detach() > 0, 1.0, 0.2)
    d = ops.conv2d_desc(2, cin, cout, h, w, kh, 4, (sh, 2), (1, 1))
    assert ops.conv2d_bwd_data_kernel_name(d) == variant
    sigma = torch.tensor([1.3]).to(DEV)
    pk = ops.conv2d_pack_bwd(d, wt.to(DEV), sigma)
    close(ops.conv2d_bwd_data(d, dy.to(DEV), pk), xin.grad, 2e-5)
    got = ops.conv2d_bwd_data(d, dy.to(DEV), pk, xin.detach().to(DEV), 0.2, add=extra.to(DEV))
    close(got, (xin.grad + extra) * slope_mask, 2e-5)
    knob(0)
    assert ops.conv2d_bwd_data_kernel_name(d).startswith("conv_mfma")
    old = ops.conv2d_bwd_data(d, dy.to(DEV), ops.conv2d_pack_bwd(d, wt.to(DEV), sigma), xin.detach().to(DEV), 0.2,
                              add=extra.to(DEV))
    close(old, (xin.grad + extra) * slope_mask, 2e-5)


def test_32_row_variant_forward_and_backward(knob):
    """32 -> 32 3x3 (first block of the STFT discriminators): 32 x 512 tiles, the chunk of the tile image copied flat."""
    torch.manual_seed(9)
    for (h, w) in [(5, 1024), (4, 400), (1, 512)]:
        pre = torch.randn(2, 32, h, w)
        xin = F.leaky_relu(pre, 0.2).detach().requires_grad_(True)
        wt = torch.randn(32, 32, 3, 3) / (32 * 9) ** 0.5
        b = torch.randn(32)
        y = F.conv2d(xin, wt, b, padding=(1, 1))
        dy = torch.randn_like(y)
        y.backward(dy)
        d = ops.conv2d_desc(2, 32, 32, h, w, 3, 3, (1, 1), (1, 1), EPI_LEAKY_PRE, 0.2)
        assert ops.conv2d_kernel_name(d) == "conv_p2d<k3,32x512>" and ops.conv2d_bwd_data_kernel_name(d) == "conv_p2d<k3,32x512>"
        close(ops.conv2d_forward(d, xin.detach().to(DEV), ops.conv2d_pack(d, wt.to(DEV)), b.to(DEV)),
              F.leaky_relu(y.detach(), 0.2), 1e-5)
        d0 = ops.conv2d_desc(2, 32, 32, h, w, 3, 3, (1, 1), (1, 1))
        got = ops.conv2d_bwd_data(d0, dy.to(DEV), ops.conv2d_pack_bwd(d0, wt.to(DEV)), xin.detach().to(DEV), 0.2)
        close(got, xin.grad * torch.where(xin.detach() > 0, 1.0, 0.2), 2e-5)
