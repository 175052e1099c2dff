"""The bf16x3 ring form of the fused residual block (csrc/resblock_b3.hip): every (C, dilation) it is instantiated for
against the oracle (fp32-class tolerance: the arithmetic is three bf16 pieces per operand, six bf16 MFMAs per product
block, fp32 accumulation -- NOT the bitwise fp32 chain) and against the fp32 ring kernel; clips shorter than one tile,
ragged lengths (any L: the input is split element by element), more tiles than workgroups (cross-tile pipeline: weight
ring, plane buffers and the next chunk's loads run over tile boundaries)."""
import pytest
import torch

from audio_generation_amd import _lib, ops
from audio_generation_amd.vae import CausalResidualBlock1d
from oracle import codec
from tests.helpers import max_abs

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 1e-5       # of max|y|: bf16x3 error vs fp64 is ~1.5e-6 of the output scale on these layers


def _block(c, d, seed):
    gen = torch.Generator().manual_seed(seed)
    sd = {}
    for name, k in (("conv1", 7), ("conv2", 1)):
        v = torch.randn(c, c, k, generator=gen) / (c * k) ** 0.5
        sd[f"{name}.conv.weight_v"] = v
        sd[f"{name}.conv.weight_g"] = v.reshape(c, -1).norm(dim=1).reshape(-1, 1, 1) * 1.1
        sd[f"{name}.conv.bias"] = torch.randn(c, generator=gen) * 0.1
    m = CausalResidualBlock1d(c, c, dilation=d)
    m.load_state_dict(sd)
    return m.to(DEV).eval(), sd, gen


def _name(m, x, impl):
    c = m.conv1.conv
    d = ops.conv_desc(_lib.CONV_CAUSAL, x.shape[0], c.in_channels, c.out_channels, x.shape[2], 7, 1, m.conv1.dilation, impl=impl)
    return ops.resblock_kernel_name(d)


def _run(m, x, impl, slope=0.1):
    m.conv1.impl = impl
    try:
        with torch.no_grad():
            return m.run(x.to(DEV), slope)
    finally:
        m.conv1.impl = _lib.IMPL_AUTO


@pytest.mark.parametrize("c", [32, 64, 128, 256])
@pytest.mark.parametrize("d", [1, 3, 9])
def test_small_ragged_and_multi_tile_clips(c, d):
    m, sd, gen = _block(c, d, 300 + c + d)
    bn = {32: 512, 64: 512, 128: 256, 256: 128}[c]
    for b, length in ((1, 3), (2, 61), (3, bn), (2, bn + 1), (1, 2 * bn + 37), (2, 1000)):
        x = torch.randn(b, c, length, generator=gen)
        want = codec.leaky(codec.residual_block(x, sd, "", d))
        assert _name(m, x, _lib.IMPL_MFMA_BF16X3).startswith("resblock_b3"), _name(m, x, _lib.IMPL_MFMA_BF16X3)
        y = _run(m, x, _lib.IMPL_MFMA_BF16X3)
        tol = TOL * max(1.0, float(want.abs().max()))
        assert max_abs(y.cpu(), want) < tol, (c, d, b, length, max_abs(y.cpu(), want))
        y32 = _run(m, x, _lib.IMPL_AUTO)
        assert max_abs(y, y32) < tol
    # without the trailing activation
    x = torch.randn(2, c, 300, generator=gen)
    m.conv1.impl = _lib.IMPL_MFMA_BF16X3
    try:
        with torch.no_grad():
            y = m(x.to(DEV))
    finally:
        m.conv1.impl = _lib.IMPL_AUTO
    assert max_abs(y.cpu(), codec.residual_block(x, sd, "", d)) < TOL * 10


@pytest.mark.parametrize("c,d,b,length", [(32, 9, 8, 36000), (64, 3, 6, 24000), (128, 1, 4, 17000), (64, 9, 2, 72000),
                                          (256, 3, 20, 3400)])
def test_more_tiles_than_workgroups(c, d, b, length):
    m, sd, gen = _block(c, d, 17 + c)
    x = torch.randn(b, c, length, generator=gen)
    bn = {32: 512, 64: 512, 128: 256, 256: 128}[c]
    assert b * -(-length // bn) > 256
    want = codec.leaky(codec.residual_block(x, sd, "", d))
    y = _run(m, x, _lib.IMPL_MFMA_BF16X3)
    assert max_abs(y.cpu(), want) < TOL * max(1.0, float(want.abs().max()))
    y2 = _run(m, x, _lib.IMPL_MFMA_BF16X3)
    assert torch.equal(y, y2)                  # run to run, bit for bit


def test_other_dilations_keep_the_first_bf16x3_kernel():
    m, sd, gen = _block(64, 2, 6)
    x = torch.randn(1, 64, 512, generator=gen)
    assert _name(m, x, _lib.IMPL_MFMA_BF16X3).startswith("resblock_mfma")
    y = _run(m, x, _lib.IMPL_MFMA_BF16X3)
    assert max_abs(y.cpu(), codec.leaky(codec.residual_block(x, sd, "", 2))) < TOL * 10
