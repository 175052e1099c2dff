"""The persistent ring form of the fused residual block (csrc/resblock_p.hip) against the oracle and against the
first kernel (csrc/resblock_mfma.hip): every (C, dilation) it is instantiated for, clips shorter than one tile,
ragged last tiles, and grids with MORE tiles than resident workgroups (the cross-tile pipeline: chunk ring, DMA cursor
and operand prefetch run over tile boundaries)."""
import pytest
import torch

from audio_generation_amd import _lib, ops
from audio_generation_amd.vae import CausalResidualBlock1d
from oracle import codec
from tests.helpers import max_abs

pytestmark = pytest.mark.gpu
DEV = "cuda"


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


def _set(knob, v):
    assert _lib.load().agx_set_tuning(knob.encode(), v) == 0


def _kernel_name(m, x):
    c = m.conv1.conv
    d = ops.conv_desc(_lib.CONV_CAUSAL, x.shape[0], c.in_channels, c.out_channels, x.shape[2], 7, 1, m.conv1.dilation)
    return ops.resblock_kernel_name(d)


@pytest.mark.parametrize("c", [32, 64, 128, 256])
@pytest.mark.parametrize("d", [1, 3, 9])
def test_small_and_ragged_clips(c, d):
    m, sd, gen = _block(c, d, 100 + c + d)
    for b, length in ((1, 4), (2, 60), (3, {32: 512, 64: 256, 128: 128, 256: 128}[c]), (2, 1000), (1, 2052)):
        x = torch.randn(b, c, length, generator=gen)
        want = codec.residual_block(x, sd, "", d)
        try:
            _set("rb_impl", 1)
            assert _kernel_name(m, x).startswith("resblock_p"), _kernel_name(m, x)
            with torch.no_grad():
                y_act = m.run(x.to(DEV), 0.1)
                y_plain = m(x.to(DEV))
            _set("rb_impl", 0)
            assert _kernel_name(m, x).startswith("resblock_mfma")
            with torch.no_grad():
                y_old = m.run(x.to(DEV), 0.1)
        finally:
            _set("rb_impl", 1)
        tol = 3e-5 * max(1.0, float(want.abs().max()))
        assert max_abs(y_act.cpu(), codec.leaky(want)) < tol, (c, d, b, length)
        assert max_abs(y_plain.cpu(), want) < tol, (c, d, b, length)
        assert max_abs(y_act, y_old) < tol


@pytest.mark.parametrize("c,d,b,length", [(32, 9, 8, 36000), (64, 3, 6, 24000), (128, 1, 4, 17000), (64, 9, 2, 72000),
                                          (256, 3, 20, 3400)])
def test_more_tiles_than_workgroups(c, d, b, length):
    """> 512 tiles: every workgroup walks over several tiles (different clips and time blocks)."""
    m, sd, gen = _block(c, d, 7 + c)
    x = torch.randn(b, c, length, generator=gen)
    bn = {32: 512, 64: 256, 128: 128, 256: 128}[c]
    assert b * -(-length // bn) > 512
    want = codec.leaky(codec.residual_block(x, sd, "", d))
    with torch.no_grad():
        y = m.run(x.to(DEV), 0.1)
    assert max_abs(y.cpu(), want) < 3e-5 * max(1.0, float(want.abs().max()))


def test_unsupported_shapes_fall_back_to_the_first_kernel():
    m, sd, gen = _block(64, 3, 5)
    x = torch.randn(1, 64, 515, generator=gen)            # length % 4 != 0
    assert _kernel_name(m, x).startswith("resblock_mfma")
    m2, sd2, _ = _block(64, 2, 6)                          # a dilation the ring kernel is not instantiated for
    x2 = torch.randn(1, 64, 512, generator=gen)
    assert _kernel_name(m2, x2).startswith("resblock_mfma")
    with torch.no_grad():
        y = m2.run(x2.to(DEV), 0.1)
    assert max_abs(y.cpu(), codec.leaky(codec.residual_block(x2, sd2, "", 2))) < 3e-5
