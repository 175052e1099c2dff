"""The persistent ring form of the resampling / stride-1 convolutions (csrc/conv_p.hip) against the oracle and the first
MFMA kernel (csrc/conv_mfma.hip): every layer geometry it is instantiated for, ragged lengths (right pad of
``_calc_extra_pad``, vae.py:39-43), clips shorter than one tile, and grids with more tiles than resident workgroups."""
import pytest
import torch

from audio_generation_amd import _lib, ops
from oracle import codec
from tests.helpers import max_abs

pytestmark = pytest.mark.gpu
DEV = "cuda"
KIND = {"conv": _lib.CONV_CAUSAL, "convt": _lib.CONV_TRANSPOSED, "upconv": _lib.CONV_UPSAMPLE}

# (variant, kind, cin, cout, k, stride): the resampling layers of config S + the two stride-1 layers around the bottleneck
LAYERS = [
    ("down2", "conv", 32, 64, 5, 2), ("down4", "conv", 64, 128, 9, 4), ("down5", "conv", 128, 256, 11, 5),
    ("down8", "conv", 256, 512, 17, 8), ("k3", "conv", 512, 512, 3, 1), ("k7", "convt", 512, 512, 7, 1),
    ("k7", "conv", 128, 256, 7, 1), ("up8", "upconv", 512, 256, 17, 8), ("up5", "upconv", 256, 128, 11, 5),
    ("up4", "upconv", 128, 64, 9, 4), ("up2", "upconv", 64, 32, 5, 2),
    # the class-default strides (2, 3, 4, 4, 5) of CausalVQAE (vae.py:215)
    ("down3", "conv", 64, 128, 7, 3), ("up3", "upconv", 128, 64, 7, 3), ("down4", "conv", 256, 512, 9, 4), ("up4", "upconv", 512, 256, 9, 4),
]


def _set(knob, v):
    assert _lib.load().agx_set_tuning(knob.encode(), v) == 0


def _case(kind, cin, cout, k, s, b, length, gen, act=True):
    wshape = (cin, cout, k) if kind == "convt" else (cout, cin, k)
    v = torch.randn(wshape, generator=gen) / (cin * k) ** 0.5
    g = torch.rand((wshape[0], 1, 1), generator=gen) + 0.5
    bias = torch.randn(cout, generator=gen) * 0.1
    x = torch.randn(b, cin, length, generator=gen)
    w = codec.fold_weight_norm(g, v)
    if kind == "conv":
        want = codec.causal_conv1d(x, w, bias, stride=s)
    elif kind == "convt":
        want = codec.causal_conv_t1d(x, w, bias, stride=s)
    else:
        want = codec.upsample_conv1d(x, w, bias, s)
    if act:
        want = codec.leaky(want)
    desc = ops.conv_desc(KIND[kind], b, cin, cout, length, k, s, 1, _lib.EPI_LEAKY_PRE if act else 0, 0.1, _lib.IMPL_AUTO)
    packed = ops.conv_pack(desc, v.to(DEV), g.to(DEV))
    return desc, packed, bias.to(DEV), x.to(DEV), want


@pytest.mark.parametrize("variant,kind,cin,cout,k,s", LAYERS)
def test_every_geometry_against_oracle_and_first_kernel(variant, kind, cin, cout, k, s):
    gen = torch.Generator().manual_seed(sum(map(ord, variant)) + cin)
    # lengths: multiples of 4 and ragged ones (16-byte cells straddling the row end are shifted into place in LDS)
    for b, length, act in ((1, 4, True), (2, 60, False), (1, 132, True), (3, 520, True), (2, 1028, False), (2, 225, True),
                           (3, 45, False), (1, 77, True), (2, 131, True), (1, 1026, True)):
        desc, packed, bias, x, want = _case(kind, cin, cout, k, s, b, length, gen, act)
        try:
            _set("conv_impl", 1)
            name = ops.conv_kernel_name(desc)
            assert name.startswith(f"conv_p<{variant},"), (name, variant)
            y = ops.conv_forward(desc, x, packed, bias)
            _set("conv_impl", 0)
            assert ops.conv_kernel_name(desc).startswith("conv_mfma")
            y_old = ops.conv_forward(desc, x, packed, bias)
        finally:
            _set("conv_impl", 1)
        tol = 2e-5 * max(1.0, float(want.abs().max()))
        assert tuple(y.shape) == tuple(want.shape)
        assert max_abs(y.cpu(), want) < tol, (variant, b, length, max_abs(y.cpu(), want))
        assert max_abs(y, y_old) < tol


@pytest.mark.parametrize("variant,kind,cin,cout,k,s,b,length", [
    ("down2", "conv", 32, 64, 5, 2, 5, 72000),        # 141 x 5 = 705 tiles of 64 x 256
    ("down4", "conv", 64, 128, 9, 4, 4, 72000),       # 141 x 4 = 564
    ("down5", "conv", 128, 256, 11, 5, 8, 24000),     # 38 x 2 x 8 = 608
    ("down8", "conv", 256, 512, 17, 8, 20, 1800),     # 4 x 4 x 20 = 320 > 256 (one workgroup per CU)
    ("up8", "upconv", 512, 256, 17, 8, 10, 228),      # 16 x 4 x 10 = 640
    ("up5", "upconv", 256, 128, 11, 5, 9, 1800),      # 5 x 15 x 9 = 675
    ("up4", "upconv", 128, 64, 9, 4, 4, 9000),        # 2 x 71 x 4 = 568
    ("up2", "upconv", 64, 32, 5, 2, 4, 36000),        # 141 x 4 = 564
    ("k3", "conv", 512, 512, 3, 1, 36, 225),          # 4 x 4 x 36 = 576; config S's bottleneck length (225 = 4 x 56 + 1)
    ("k7", "convt", 512, 512, 7, 1, 36, 225),
    ("up8", "upconv", 512, 256, 17, 8, 9, 225),
])
def test_more_tiles_than_workgroups(variant, kind, cin, cout, k, s, b, length):
    gen = torch.Generator().manual_seed(3)
    desc, packed, bias, x, want = _case(kind, cin, cout, k, s, b, length, gen)
    assert ops.conv_kernel_name(desc).startswith(f"conv_p<{variant},")
    y = ops.conv_forward(desc, x, packed, bias)
    assert max_abs(y.cpu(), want) < 2e-5 * max(1.0, float(want.abs().max()))


def test_calls_the_ring_kernel_does_not_cover_fall_back():
    gen = torch.Generator().manual_seed(4)
    desc, packed, bias, x, want = _case("conv", 64, 128, 9, 4, 2, 3, gen)         # shorter than one 16-byte cell
    assert ops.conv_kernel_name(desc).startswith("conv_mfma")
    assert max_abs(ops.conv_forward(desc, x, packed, bias).cpu(), want) < 2e-5 * max(1.0, float(want.abs().max()))
    # a second LeakyReLU behind the residual needs a residual; the upsampling (multi-phase) geometries only take bias + LeakyReLU
    d2 = ops.conv_desc(_lib.CONV_UPSAMPLE, 2, 512, 256, 64, 17, 8, 1, _lib.EPI_RESIDUAL, 0.1, _lib.IMPL_AUTO)
    assert ops.conv_kernel_name(d2).startswith("conv_mfma")


# ---- round 4: the one-phase geometries behind configs 3 / 4 (every Linear of the transformer block as a k = 1 conv,
# transformers.py:157-223; WaveletLayer's two padding="same" convs, wavelets.py:193-201) and their epilogues ----------------
def _ref_epilogue(pre, epi, res):
    import torch.nn.functional as F
    v = pre
    if epi & _lib.EPI_LEAKY_PRE:
        v = codec.leaky(v)
    if epi & _lib.EPI_GELU_PRE:
        v = F.gelu(v)
    if epi & _lib.EPI_RESIDUAL:
        v = v + res
    if epi & _lib.EPI_LEAKY_POST:
        v = codec.leaky(v)
    return v


@pytest.mark.parametrize("variant,kind,cin,cout,k", [
    ("k1", _lib.CONV_CAUSAL, 512, 512, 1), ("k1", _lib.CONV_CAUSAL, 512, 1536, 1), ("k1", _lib.CONV_CAUSAL, 256, 256, 1),
    ("k1", _lib.CONV_CAUSAL, 128, 128, 1), ("same11", _lib.CONV_SAME, 256, 512, 11), ("same3", _lib.CONV_SAME, 512, 128, 3),
    ("k3", _lib.CONV_CAUSAL, 512, 512, 3)])
def test_one_phase_geometries_and_their_epilogues(variant, kind, cin, cout, k):
    import torch.nn.functional as F
    gen = torch.Generator().manual_seed(sum(map(ord, variant)) + cin + cout)
    epis = (0, _lib.EPI_LEAKY_PRE, _lib.EPI_GELU_PRE, _lib.EPI_RESIDUAL, _lib.EPI_RESIDUAL | _lib.EPI_LEAKY_POST,
            _lib.EPI_LEAKY_PRE | _lib.EPI_RESIDUAL | _lib.EPI_LEAKY_POST)
    for n, (b, length) in enumerate(((2, 225), (1, 4), (3, 60), (2, 1028), (1, 77), (5, 131), (2, 520))):
        epi = epis[n % len(epis)]
        v = torch.randn(cout, cin, k, generator=gen) / (cin * k) ** 0.5
        g = torch.rand(cout, 1, 1, generator=gen) + 0.5
        bias = torch.randn(cout, generator=gen) * 0.1
        x = torch.randn(b, cin, length, generator=gen)
        res = torch.randn(b, cout, length, generator=gen)
        w = codec.fold_weight_norm(g, v)
        pre = codec.causal_conv1d(x, w, bias, stride=1) if kind == _lib.CONV_CAUSAL else F.conv1d(x, w, bias, padding="same")
        want = _ref_epilogue(pre, epi, res)
        desc = ops.conv_desc(kind, b, cin, cout, length, k, 1, 1, epi, 0.1, _lib.IMPL_AUTO)
        packed = ops.conv_pack(desc, v.to(DEV), g.to(DEV))
        r = res.to(DEV) if epi & _lib.EPI_RESIDUAL else None
        try:
            _set("conv_impl", 1)
            assert ops.conv_kernel_name(desc).startswith(f"conv_p<{variant},"), (ops.conv_kernel_name(desc), variant)
            y = ops.conv_forward(desc, x.to(DEV), packed, bias.to(DEV), res=r)
            _set("conv_impl", 0)
            assert not ops.conv_kernel_name(desc).startswith("conv_p")
            y_old = ops.conv_forward(desc, x.to(DEV), packed, bias.to(DEV), res=r)
        finally:
            _set("conv_impl", 1)
        tol = 2e-5 * max(1.0, float(want.abs().max()))
        assert max_abs(y.cpu(), want) < tol, (variant, epi, b, length, max_abs(y.cpu(), want))
        assert max_abs(y, y_old) < tol


@pytest.mark.parametrize("variant,kind,cin,cout,k,b,length,epi", [
    ("k1", _lib.CONV_CAUSAL, 512, 512, 1, 40, 225, _lib.EPI_RESIDUAL),            # 4 x 4 x 40 = 640 tiles of 128 x 64
    ("k1", _lib.CONV_CAUSAL, 512, 1536, 1, 32, 225, 0),                           # config 3's QKV projection: 12 x 4 x 32
    ("k1", _lib.CONV_CAUSAL, 512, 512, 1, 32, 225, _lib.EPI_GELU_PRE),            # ... FFN-in
    ("same11", _lib.CONV_SAME, 256, 512, 11, 3, 3600, 0),                         # config 4's wavelet block: 4 x 29 x 3
    ("same3", _lib.CONV_SAME, 512, 128, 3, 3, 18000, 0),                          # 282 x 3 = 846
    ("k1", _lib.CONV_CAUSAL, 128, 128, 1, 4, 9000, _lib.EPI_RESIDUAL | _lib.EPI_LEAKY_POST),   # the unfused block's second conv
])
def test_one_phase_geometries_with_more_tiles_than_workgroups(variant, kind, cin, cout, k, b, length, epi):
    import torch.nn.functional as F
    gen = torch.Generator().manual_seed(5)
    v = torch.randn(cout, cin, k, generator=gen) / (cin * k) ** 0.5
    bias = torch.randn(cout, generator=gen) * 0.1
    x = torch.randn(b, cin, length, generator=gen)
    res = torch.randn(b, cout, length, generator=gen)
    pre = codec.causal_conv1d(x, v, bias, stride=1) if kind == _lib.CONV_CAUSAL else F.conv1d(x, v, bias, padding="same")
    want = _ref_epilogue(pre, epi, res)
    desc = ops.conv_desc(kind, b, cin, cout, length, k, 1, 1, epi, 0.1, _lib.IMPL_AUTO)
    assert ops.conv_kernel_name(desc).startswith(f"conv_p<{variant},")
    y = ops.conv_forward(desc, x.to(DEV), ops.conv_pack(desc, v.to(DEV)), bias.to(DEV),
                         res=res.to(DEV) if epi & _lib.EPI_RESIDUAL else None)
    assert max_abs(y.cpu(), want) < 2e-5 * max(1.0, float(want.abs().max()))
