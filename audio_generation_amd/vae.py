"""Drop-in ``CausalVQAE`` and its causal conv modules, executed by libagx.

Same class names, constructor arguments, attribute names, forward signatures
and ``state_dict()`` keys as the reference's ``networks/vae.py`` (cited per
class), so the reference's training / sampling callers (``training.py:325-328,
488-500, 502-516``; ``utils.py:238-259``) keep working -- but no ATen conv is
called: every layer runs as a hand-written HIP kernel behind the C ABI of
``include/agx.h``.  The module tree exists for parameters and checkpoints;
``forward`` walks it and issues fused launches (residual block = one call,
activation / padding / crop / upsample folded into the conv kernels).

Differentiation: forward AND backward run on libagx (``native_backward.py``: conv
backward-data / weight-gradient kernels incl. the ``depthwise=True`` block variant, the RVQ
straight-through pass).  No ATen arithmetic op is differentiated anywhere: a configuration
without backward kernels (an activation the kernels do not fuse, ``norm != Identity``)
raises ``AgxError`` / ``NotImplementedError`` instead of falling back.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence

import torch
from torch import nn

from . import ops
from ._lib import (CONV_CAUSAL, CONV_PADDED, CONV_TRANSPOSED, CONV_UPSAMPLE, EPI_LEAKY_POST, EPI_LEAKY_PRE,
                   IMPL_AUTO, IMPL_MFMA_BF16X3, AgxError, needs_grad)

Tensor = torch.Tensor


def tuple_checker(item, length):
    """Same contract as the reference helper (networks/utils.py:212-220; also
    re-exported by the external quantiser module, vae.py:6)."""
    if isinstance(item, (int, float, str)):
        item = [item] * length
    elif isinstance(item, (tuple, list)):
        assert len(item) == length, f"Expected tuple of length {length}, got {len(item)}"
    return item


def _leaky_slope(act: Optional[nn.Module]) -> Optional[float]:
    """Negative slope if ``act`` is an activation the kernels fuse."""
    if act is None or isinstance(act, nn.Identity):
        return None
    if isinstance(act, nn.LeakyReLU):
        return float(act.negative_slope)
    if isinstance(act, nn.ReLU):
        return 0.0
    raise NotImplementedError(
        f"activation {type(act).__name__} has no HIP kernel (LeakyReLU / ReLU are fused into the convs)")


class _ConvParams(nn.Module):
    """Parameter holder standing where the reference has a weight-normed
    ``torch.nn.Conv1d`` / ``ConvTranspose1d`` (``add_util_norm``, utils.py:34-42):
    parameters ``weight_g`` (dim0,1,1), ``weight_v`` and ``bias`` under the same
    names, initialised the way torch initialises a conv + ``weight_norm``.
    With ``norm != "weight"`` it holds a plain ``weight``.
    """

    def __init__(self, c_in: int, c_out: int, kernel: int, stride: int, dilation: int, bias: bool,
                 transposed: bool, norm: str = "weight", groups: int = 1):
        super().__init__()
        self.in_channels, self.out_channels = c_in, c_out
        self.kernel_size, self.stride, self.dilation = (kernel,), (stride,), (dilation,)
        self.transposed = transposed
        self.groups = groups
        shape = (c_in, c_out, kernel) if transposed else (c_out, c_in // groups, kernel)
        w = torch.empty(shape)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        fan_in = shape[1] * kernel
        bias_p = None
        if bias:
            bound = 1.0 / math.sqrt(fan_in)
            bias_p = nn.Parameter(torch.empty(c_out).uniform_(-bound, bound))
        if norm == "weight":
            # registration order of torch's weight_norm(Conv1d): bias, weight_g, weight_v -- ``parameters()`` order is
            # what an index-based optimizer state of the reference (trainer_state.pkl, training.py:225-242) refers to
            self.register_parameter("bias", bias_p)
            self.weight_g = nn.Parameter(w.reshape(shape[0], -1).norm(dim=1).reshape(-1, 1, 1))
            self.weight_v = nn.Parameter(w)
        elif norm == "spectral":
            raise NotImplementedError("spectral norm is only used by the discriminators (out of scope)")
        else:
            self.weight = nn.Parameter(w)
            self.register_parameter("bias", bias_p)
        self._packed: Optional[Tensor] = None
        self._packed_key = None

    def packed_bwd(self, kind: int) -> Tensor:
        """Packed image of the layer's backward-data op (same invalidation rule as ``packed``)."""
        if hasattr(self, "weight_v"):
            v, g = self.weight_v, self.weight_g
            key = (kind, v.data_ptr(), v._version, g.data_ptr(), g._version)
        else:
            v, g = self.weight, None
            key = (kind, v.data_ptr(), v._version)
        if getattr(self, "_packed_bwd", None) is None or self._packed_bwd_key != key:
            desc = ops.conv_desc(kind, 1, self.in_channels, self.out_channels, 1 << 20,
                                 self.kernel_size[0], self.stride[0], self.dilation[0])
            self._packed_bwd = ops.conv_pack_bwd(desc, v.detach(), None if g is None else g.detach())
            self._packed_bwd_key = key
        return self._packed_bwd

    def packed(self, kind: int, impl: int = IMPL_AUTO) -> Tensor:
        """Packed (weight-norm folded) image, rebuilt when the parameters change
        (optimizer step, ``load_state_dict``, ``.to(device)``).  The bf16x3 kernels have their own image."""
        bf = impl == IMPL_MFMA_BF16X3
        if hasattr(self, "weight_v"):
            v, g = self.weight_v, self.weight_g
            key = (kind, bf, v.data_ptr(), v._version, g.data_ptr(), g._version)
        else:
            v, g = self.weight, None
            key = (kind, bf, v.data_ptr(), v._version)
        if self._packed is None or self._packed_key != key:
            desc = ops.conv_desc(kind, 1, self.in_channels, self.out_channels, 1 << 20,
                                 self.kernel_size[0], self.stride[0], self.dilation[0],
                                 impl=IMPL_MFMA_BF16X3 if bf else IMPL_AUTO, groups=self.groups)
            self._packed = ops.conv_pack(desc, v.detach(), None if g is None else g.detach())
            self._packed_key = key
        return self._packed


class _ConvBase(nn.Module):
    kind = CONV_CAUSAL
    impl = IMPL_AUTO  # tests override to pin a kernel family

    def run(self, x: Tensor, epilogue: int = 0, slope: float = 0.1, res: Optional[Tensor] = None) -> Tensor:
        c = self.conv
        if x.dim() != 3 or x.shape[1] != c.in_channels:
            raise AgxError(f"{type(self).__name__}: expected (B,{c.in_channels},L), got {tuple(x.shape)}")
        desc = ops.conv_desc(self.kind, x.shape[0], c.in_channels, c.out_channels, x.shape[2],
                             c.kernel_size[0], c.stride[0], c.dilation[0], epilogue, slope, self.impl,
                             groups=getattr(c, "groups", 1))
        bias = None if c.bias is None else c.bias.detach()
        return ops.conv_forward(desc, x, c.packed(self.kind, self.impl), bias, res)

    def forward(self, x: Tensor) -> Tensor:
        return self.run(x)


class CausalConv1d(_ConvBase):
    """networks/vae.py:14-43."""
    kind = CONV_CAUSAL

    def __init__(self, in_channels, out_channels, kernel_size, dilation=1, stride=1, bias=True,
                 groups=1, norm="weight"):
        super().__init__()
        if groups != 1:
            # the one grouped use in the reference: the k = 1 depthwise conv of vae.py:103 -- no padding at all,
            # so it is the grouped AGX_CONV_PADDED layer (direct kernel)
            if kernel_size != 1 or stride != 1 or dilation != 1:
                raise NotImplementedError("grouped causal convs are wired for kernel 1 / stride 1 only (vae.py:103)")
            self.kind = CONV_PADDED
        self.conv = _ConvParams(in_channels, out_channels, kernel_size, stride, dilation, bias, False, norm, groups)
        self.dilation = dilation
        self.pad = dilation * (kernel_size - 1) - stride + 1  # vae.py:32


class CausalConvT1d(_ConvBase):
    """networks/vae.py:45-64."""
    kind = CONV_TRANSPOSED

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, bias=True, norm="weight"):
        super().__init__()
        self.conv = _ConvParams(in_channels, out_channels, kernel_size, stride, 1, bias, True, norm)
        self.right_pad = kernel_size - stride


class CausalUpsampleConv1d(_ConvBase):
    """networks/vae.py:66-89 (nearest upsample + ``padding="same"`` conv; not
    causal in the reference either).  Runs as ``stride`` polyphase 3-tap filters
    on the low-rate signal -- the upsampled tensor is never materialised."""
    kind = CONV_UPSAMPLE

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, bias=True, norm="weight"):
        super().__init__()
        self.scale_factor = stride
        self.conv = _ConvParams(in_channels, out_channels, kernel_size, stride, 1, bias, False, norm)


class CausalResidualBlock1d(nn.Module):
    """networks/vae.py:91-117: ``x + conv_k1(act(conv_k7,dil(x)))``."""
    # profiling aid: issue the two convs as separate C-ABI calls (same kernels) so a
    # launch observer can time them one by one
    split_launches = False

    def __init__(self, in_channels, out_channels, kernel_size=7, dilation=1, bias=True,
                 activation=None, dropout=0.0, depthwise=False):
        super().__init__()
        if dropout != 0.0:
            raise NotImplementedError("dropout > 0 is training-only and not on the forward path")
        if in_channels != out_channels:
            raise AgxError("residual block needs in_channels == out_channels (as the reference's add does)")
        self.depthwise = depthwise
        if depthwise:   # vae.py:103-105: a per-channel k = 1 conv in front of the dilated conv (three launches, unfused)
            self.conv1 = nn.Sequential(CausalConv1d(in_channels, in_channels, 1, bias=bias, groups=in_channels),
                                       CausalConv1d(in_channels, out_channels, kernel_size, dilation=dilation, bias=bias))
            object.__setattr__(self.conv1[0], "_parent_block", self)
        else:
            self.conv1 = CausalConv1d(in_channels, out_channels, kernel_size, dilation=dilation, bias=bias)
            object.__setattr__(self.conv1, "_parent_block", self)   # plain attribute: no module cycle
        self.conv2 = CausalConv1d(out_channels, out_channels, 1, bias=bias)
        self.activation = nn.LeakyReLU(0.1) if activation is None else activation
        self.dropout = nn.Dropout(dropout)

    def run(self, x: Tensor, post_slope: Optional[float] = None) -> Tensor:
        """Whole block (+ the activation that follows it in the enclosing
        ``Sequential`` when ``post_slope`` is given) through ``agx_resblock_forward``."""
        slope = _leaky_slope(self.activation)
        if self.depthwise:
            h = self.conv1[1].run(self.conv1[0].run(x), EPI_LEAKY_PRE if slope is not None else 0, slope or 0.0)
            if slope is None:
                h = self.activation(h)
            epi = ops.EPI_RESIDUAL | (EPI_LEAKY_POST if post_slope is not None else 0)
            return self.conv2.run(h, epi, post_slope or 0.0, res=x)
        c1, c2 = self.conv1.conv, self.conv2.conv
        if self.split_launches or slope is None or (post_slope is not None and post_slope != slope):
            # exotic activation mix: two convs with separate epilogues
            h = self.conv1.run(x, EPI_LEAKY_PRE if slope is not None else 0, slope or 0.0)
            epi = ops.EPI_RESIDUAL | (EPI_LEAKY_POST if post_slope is not None else 0)
            return self.conv2.run(h, epi, post_slope or 0.0, res=x)
        desc = ops.conv_desc(CONV_CAUSAL, x.shape[0], c1.in_channels, c1.out_channels, x.shape[2],
                             c1.kernel_size[0], 1, c1.dilation[0], 0, slope, self.conv1.impl)
        b1 = None if c1.bias is None else c1.bias.detach()
        b2 = None if c2.bias is None else c2.bias.detach()
        impl = self.conv1.impl
        return ops.resblock_forward(desc, x, c1.packed(CONV_CAUSAL, impl), b1, c2.packed(CONV_CAUSAL, impl), b2,
                                    post_act=post_slope is not None)

    def forward(self, x: Tensor) -> Tensor:
        return self.run(x, None)


def _default_act():
    return nn.LeakyReLU(0.1)


class CausalEncoderBlock(nn.Module):
    """networks/vae.py:119-148."""

    def __init__(self, in_channels, out_channels, stride, n_layers=4, activation=None, depthwise=False, multires=None):
        super().__init__()
        activation = _default_act() if activation is None else activation
        layers = [nn.Sequential(CausalResidualBlock1d(in_channels, in_channels, dilation=3 ** i,
                                                      depthwise=depthwise), activation)
                  for i in range(n_layers - 1)]
        # multires = (kernel_size, depth): BUILD-DEFINED placement of CausalMultiresConv1d (the reference imports it at vae.py:7
        # and never wires it) -- right behind the strided conv, its GELU in place of the block's activation there
        layers.append(nn.Sequential(CausalConv1d(in_channels, out_channels, 2 * stride + 1, stride=stride),
                                    nn.Identity() if multires else activation))
        self.layers = nn.ModuleList(layers)
        if multires:
            from .wavelets import CausalMultiresConv1d
            self.multires = CausalMultiresConv1d(out_channels, multires[0], multires[1])

    def forward(self, x: Tensor) -> Tensor:
        for seq in self.layers:
            x = _run_fused_pair(seq[0], seq[1], x)
        if hasattr(self, "multires"):
            x = self.multires._hip(x)
        return x


class CausalDecoderBlock(nn.Module):
    """networks/vae.py:150-202."""

    def __init__(self, in_channels, out_channels, stride, n_layers=4, activation=None, depthwise=False,
                 upsample=True, wavelet=False, wavelet_hidden_ratio=4, channelwise=True, multires=None):
        super().__init__()
        activation = _default_act() if activation is None else activation
        self.wavelet = wavelet
        if wavelet:
            from .wavelets import WaveletLayer
            conv_layer = WaveletLayer(in_channels, out_channels * wavelet_hidden_ratio,
                                      out_channels=out_channels, scale_factor=stride,
                                      wavelet_kernel_size=2 * stride + 1,
                                      n_points=2 * stride * wavelet_hidden_ratio,
                                      channelwise_scale=channelwise)
        elif upsample:
            conv_layer = CausalUpsampleConv1d(in_channels, out_channels, 2 * stride + 1, stride=stride)
        else:
            conv_layer = CausalConvT1d(in_channels, out_channels, 2 * stride + 1, stride=stride)
        self.in_conv = nn.Sequential(conv_layer, nn.Identity() if multires else activation)
        if multires:   # build-defined placement, as in CausalEncoderBlock: behind the resampling layer, GELU instead of the activation
            from .wavelets import CausalMultiresConv1d
            self.multires = CausalMultiresConv1d(out_channels, multires[0], multires[1])
        self.layers = nn.ModuleList([
            nn.Sequential(CausalResidualBlock1d(out_channels, out_channels, dilation=3 ** i,
                                                depthwise=depthwise), activation)
            for i in range(n_layers - 1)])

    def forward(self, x: Tensor) -> Tensor:
        x = _run_fused_pair(self.in_conv[0], self.in_conv[1], x)
        if hasattr(self, "multires"):
            x = self.multires._hip(x)
        for seq in self.layers:
            x = _run_fused_pair(seq[0], seq[1], x)
        return x


def _run_unit_forward(unit, x: Tensor) -> Tensor:
    """Forward of one unit of native_backward.build_units (same fused kernels as inference)."""
    if unit.kind in ("res", "resdw"):
        res = unit.convs[0]._parent_block
        return res.run(x, unit.slope)
    if unit.kind == "wavelet":
        return unit.convs[0].run_fused(x, unit.slope)
    if unit.kind == "multires":
        return unit.convs[0]._hip(x)
    conv = unit.convs[0]
    return conv.run(x, EPI_LEAKY_PRE if unit.slope is not None else 0, unit.slope or 0.0)


def _run_fused_pair(layer: nn.Module, act: nn.Module, x: Tensor) -> Tensor:
    """``act(layer(x))`` with the activation folded into the layer's kernel."""
    slope = _leaky_slope(act)
    if isinstance(layer, CausalResidualBlock1d):
        return layer.run(x, slope)
    if isinstance(layer, _ConvBase):
        return layer.run(x, EPI_LEAKY_PRE if slope is not None else 0, slope or 0.0)
    if hasattr(layer, "run_fused"):  # WaveletLayer
        return layer.run_fused(x, slope)
    raise NotImplementedError(f"no HIP path for {type(layer).__name__}")


class CausalVQAE(nn.Module):
    """networks/vae.py:204-351 -- same constructor, attributes and methods."""

    def __init__(self, in_channels=1, n_blocks=5, n_layers_per_block=4, first_block_channels=32,
                 num_quantizers=8, codebook_size=1024, codebook_dim=512, vq_cutoff_freq=1,
                 vq_type="ema", strides=(2, 3, 4, 4, 5), input_format="b l c", channel_multiplier=2,
                 norm=nn.Identity, depthwise=False, use_som=True, som_kernel_type="hard",
                 wavelet_decoders=(False, True, False, False, False),
                 multires_encoders=False, multires_decoders=False, multires_kernel_size=2, multires_depth=3):
        """The last four arguments are BUILD-DEFINED (default off: the reference's model): block i gets a
        ``CausalMultiresConv1d(channels, multires_kernel_size, multires_depth)`` (wavelets.py:38-96 -- imported by the
        reference at vae.py:7 and never wired) right behind its strided / upsampling conv, whose GELU takes the place of the block
        activation there; lists are in encoder order resp. decoder order.  BASELINE config 4 ("multiresolution / wavelet layers in
        encoder + decoder") at model level; oracle: oracle/codec.py CodecSpec."""
        super().__init__()
        from .quantizer import ResidualQuantizer

        self.in_channels = in_channels
        self.n_blocks = n_blocks
        self.n_layers_per_block = n_layers_per_block
        self.num_quantizers = num_quantizers
        self.vq_cutoff_freq = vq_cutoff_freq
        self.codebook_dim = codebook_dim
        self.codebook_size = tuple_checker(codebook_size, num_quantizers)
        self.strides = tuple_checker(strides, n_blocks)
        self.scale_factor = int(math.prod(int(s) for s in self.strides))
        self.input_format = input_format

        if isinstance(wavelet_decoders, (list, tuple)):
            assert len(wavelet_decoders) == n_blocks, "Number of wavelet decoders must match number of blocks."
            self.wavelet_decoders = list(wavelet_decoders)[::-1]  # vae.py:240: iterated backwards
        else:
            self.wavelet_decoders = [wavelet_decoders] * n_blocks

        self.quantizer = ResidualQuantizer(num_quantizers=num_quantizers, dim=codebook_dim,
                                           quantizer_class=vq_type, codebook_sizes=codebook_size,
                                           vq_cutoff_freq=vq_cutoff_freq, use_som=use_som,
                                           som_kernel_type=som_kernel_type)

        self.multires_encoders = list(tuple_checker(multires_encoders, n_blocks))
        self.multires_decoders = list(tuple_checker(multires_decoders, n_blocks))
        mr = (multires_kernel_size, multires_depth)
        ch = [first_block_channels * channel_multiplier ** i for i in range(n_blocks + 1)]
        encoders: List[nn.Module] = [nn.Sequential(norm(), CausalConv1d(in_channels, first_block_channels, 7))]
        for i in range(n_blocks):
            encoders.append(CausalEncoderBlock(ch[i], ch[i + 1], self.strides[i], n_layers_per_block,
                                               depthwise=depthwise, multires=mr if self.multires_encoders[i] else None))
        encoders.append(CausalConv1d(ch[-1], codebook_dim, 3))

        decoders: List[nn.Module] = [CausalConvT1d(codebook_dim, ch[-1], 7)]
        for i in range(n_blocks, 0, -1):
            decoders.append(CausalDecoderBlock(ch[i], ch[i - 1], self.strides[i - 1],
                                               n_layers=n_layers_per_block, depthwise=depthwise,
                                               wavelet=self.wavelet_decoders[i - 1],
                                               multires=mr if self.multires_decoders[n_blocks - i] else None))
        decoders.append(CausalConv1d(first_block_channels, in_channels, 7))

        self.encoders = nn.ModuleList(encoders)
        self.decoders = nn.ModuleList(decoders)

    # -- arithmetic of the conv stacks (build-defined; the reference has fp32 only) ------------------
    def set_conv_arithmetic(self, decoders: str = "fp32", encoders: str = "fp32") -> "CausalVQAE":
        """``"fp32"``: the fp32-input MFMA kernels (bitwise an fp32 FMA chain).  ``"bf16x3"``: operands split into three
        bf16 pieces on the bf16 MFMA -- the same error against fp64 as fp32 (tools/bf16x3_accuracy.py), 1.3-1.5x
        faster, but not the bitwise fp32 chain.  The default keeps the ENCODER exact (it decides the RVQ indices)
        and is what every parity statement refers to; ``bench.py`` reports which setting it ran."""
        for stack, mode in ((self.decoders, decoders), (self.encoders, encoders)):
            if mode not in ("fp32", "bf16x3"):
                raise ValueError(f"unknown arithmetic {mode!r}")
            in_block = {id(c_) for blk in stack.modules() if isinstance(blk, CausalResidualBlock1d)
                        for c_ in blk.modules() if isinstance(c_, _ConvBase)}
            for m in stack.modules():
                if isinstance(m, _ConvBase):
                    c = m.conv
                    q = c.stride[0] if m.kind in (CONV_TRANSPOSED, CONV_UPSAMPLE) else 1
                    ok = mode == "bf16x3" and c.in_channels % 16 == 0 and q * c.out_channels >= 32 and getattr(c, "groups", 1) == 1
                    # resampling convs: bf16x3 where the layer has the ring form (conv_b3.hip: the decoder's up-convs and
                    # the k = 7 transposed conv) or where the first-round bf16x3 kernel beats the fp32 ring kernel (the
                    # stride-4 down-conv; tools/layer_times.py bf16x3); the others stay on the fp32 ring, which is at least as exact
                    if ok and id(m) not in in_block:
                        nominal = ops.conv_desc(m.kind, 1, c.in_channels, c.out_channels, 4096, c.kernel_size[0], c.stride[0],
                                                c.dilation[0], EPI_LEAKY_PRE, 0.1, IMPL_MFMA_BF16X3)
                        ok = ops.conv_kernel_name(nominal).startswith("conv_b3") or (m.kind == CONV_CAUSAL and c.stride[0] == 4)
                    m.impl = IMPL_MFMA_BF16X3 if ok else IMPL_AUTO
        self.__dict__.pop("_unit_cache", None)
        return self

    # -- layout helpers (vae.py:283-288) ------------------------------------------------
    def rearrange_in(self, x: Tensor) -> Tensor:
        return x.transpose(1, 2).contiguous() if self.input_format == "b l c" else x

    def rearrange_out(self, x: Tensor) -> Tensor:
        return x.transpose(1, 2).contiguous() if self.input_format == "b l c" else x

    def _encoders_hip(self, x: Tensor) -> Tensor:
        first = self.encoders[0]
        if not isinstance(first[0], nn.Identity):
            raise NotImplementedError("only norm=Identity (the reference default) has a HIP path")
        x = first[1].run(x)
        for enc in list(self.encoders)[1:]:
            x = enc(x)
        return x

    def _decoders_hip(self, x: Tensor) -> Tensor:
        decs = list(self.decoders)
        x, decs = self._decoder_head_on_planes(x, decs)
        for dec in decs:
            x = dec(x)
        return x

    def _decoder_head_on_planes(self, x: Tensor, decs):
        """bf16x3 decoders, inference: the k = 7 transposed conv (vae.py:269) writes its output as ACTIVATION PLANES
        (include/agx.h: the three bf16 pieces of every element, split once in the producer's epilogue) and the first
        block's polyphase up-conv (vae.py:176-179; M = 8 x 256 rows = 16 row blocks that each re-split the same input
        tile otherwise) stages them by LDS-DMA.  Bit-identical to the fp32-activation path (same pieces, same products,
        same order: tests/test_gpu_conv_b3.py, test_gpu_fullsize.py); any other configuration takes the plain path."""
        if torch.is_grad_enabled() or len(decs) < 2 or x.dim() != 3:
            return x, decs
        head, blk = decs[0], decs[1]
        if not (isinstance(head, CausalConvT1d) and head.impl == IMPL_MFMA_BF16X3 and isinstance(blk, CausalDecoderBlock)
                and not blk.wavelet and not hasattr(blk, "multires")):
            return x, decs
        up, act = blk.in_conv[0], blk.in_conv[1]
        slope = _leaky_slope(act)
        if not (isinstance(up, CausalUpsampleConv1d) and up.impl == IMPL_MFMA_BF16X3 and x.shape[1] % 8 == 0):
            return x, decs
        hc, uc = head.conv, up.conv
        d0 = ops.conv_desc(head.kind, x.shape[0], hc.in_channels, hc.out_channels, x.shape[2], hc.kernel_size[0], hc.stride[0],
                           hc.dilation[0], 0, 0.1, head.impl)
        if hc.stride[0] != 1 or ops.conv_planes_supported(d0) != 2:
            return x, decs
        d1 = ops.conv_desc(up.kind, x.shape[0], uc.in_channels, uc.out_channels, ops.conv_out_len(d0), uc.kernel_size[0], uc.stride[0],
                           uc.dilation[0], EPI_LEAKY_PRE if slope is not None else 0, slope or 0.0, up.impl)
        if ops.conv_planes_supported(d1) < 1:
            return x, decs
        xp = ops.planes_split(x)                                             # the RVQ's output: 15 MB at config S
        yp = ops.conv_forward_planes(d0, xp, hc.packed(head.kind, head.impl), None if hc.bias is None else hc.bias.detach(),
                                     out_planes=True)
        h = ops.conv_forward_planes(d1, yp, uc.packed(up.kind, up.impl), None if uc.bias is None else uc.bias.detach())
        for seq in blk.layers:
            h = _run_fused_pair(seq[0], seq[1], h)
        return h, decs[2:]

    def _run_encoders(self, x: Tensor) -> Tensor:
        """Encoder stack; when a gradient is needed, forward + backward on the HIP kernels (native_backward.py)."""
        if needs_grad(x, self.encoders):
            from .native_backward import run_stack
            return run_stack(self._units("encoders"), x)
        return self._encoders_hip(x)

    def _run_decoders(self, x: Tensor) -> Tensor:
        if needs_grad(x, self.decoders):
            from .native_backward import run_stack
            return run_stack(self._units("decoders"), x)
        return self._decoders_hip(x)

    def _units(self, which: str):
        """Flattened unit list of a stack for the native backward; a stack with a layer the backward kernels do not
        cover raises (there is no ATen fallback)."""
        cache = self.__dict__.setdefault("_unit_cache", {})
        if which not in cache:
            from .native_backward import build_units
            if which == "encoders" and not isinstance(self.encoders[0][0], nn.Identity):
                raise NotImplementedError("only norm=Identity (the reference default) has HIP kernels")
            cache[which] = build_units(getattr(self, which))
        return cache[which]

    def encode(self, x, update_codebook=False, codebook_n=None, prioritize_early=False):
        """vae.py:307-322 -> (x_q (B,C,T), commit_loss, index (B,T,Q))."""
        z = self._run_encoders(self.rearrange_in(x))  # (B, D, T)
        q = self.quantizer
        if hasattr(q, "quantize_bcl"):
            # native quantiser: reads and writes the (B,D,T) layout directly, no transposes
            zq, index, commit = q.quantize_bcl(z, codebook_n, update_codebook=update_codebook,
                                               prioritize_early=prioritize_early)
        else:
            # any replacement bottleneck honouring the reference call contract (vae.py:315-318)
            zq, index, commit = q(z.transpose(1, 2), codebook_n, update_codebook=update_codebook,
                                  prioritize_early=prioritize_early)
            zq = zq.transpose(1, 2).contiguous()
        return zq, commit, index

    def forward(self, x, update_codebook=False, codebook_n=None, prioritize_early=False):
        """vae.py:293-305 -> (y, commit_loss, index)."""
        zq, commit, index = self.encode(x, update_codebook, codebook_n, prioritize_early)
        y = self.rearrange_out(self._run_decoders(zq))
        return y, commit, index

    def decode(self, zq: Tensor) -> Tensor:
        """Decoder half of ``forward`` on (B, D, T) quantised latents."""
        return self.rearrange_out(self._run_decoders(zq))

    def sample(self, length=225, device="cuda", normal_var=5e3, n_iters=12):
        """vae.py:324-345: random codes -> dequantise -> decode."""
        self.to(device)
        was_training = self.training
        self.eval()
        with torch.no_grad():
            x = None
            for i in range(self.num_quantizers):
                idx = torch.randint(0, self.codebook_size[0], (1, length), device=device)
                x_i = self.quantizer.quantizers[i].dequantize(idx)
                x = x_i if x is None else x + x_i
            y = self.rearrange_out(self._run_decoders(x.transpose(1, 2).contiguous()))
        self.train(True)  # the reference unconditionally returns to train mode (vae.py:344)
        del was_training
        return y

    # -- codec wire format (bit budget of utils.py:137-147) ---------------------
    def compress(self, x, codebook_n=None):
        """Waveform -> (uint8 bitstream, (B, T, Q)).  ceil(log2(K)) bits per code, dense."""
        with torch.no_grad():
            _, _, index = self.encode(x, codebook_n=codebook_n)
        bits = max(1, (max(int(k) for k in self.codebook_size) - 1).bit_length())   # every stage at the widest stage's width
        return ops.codes_pack(index, bits), tuple(index.shape)

    def decompress(self, stream, shape):
        """Inverse of ``compress``: bitstream -> codes -> sum of codewords -> decoder."""
        b, t, q = shape
        bits = max(1, (max(int(k) for k in self.codebook_size) - 1).bit_length())   # every stage at the widest stage's width
        index = ops.codes_unpack(stream, b * t * q, bits).reshape(b, t, q)
        with torch.no_grad():
            zq = None
            for i in range(q):
                zq = ops.rvq_dequantize(self.quantizer.codebooks.detach()[i], index[..., i], out=zq,
                                        accumulate=zq is not None)
            return self.decode(zq.transpose(1, 2).contiguous()), index

    def replace_quantizer(self, new_quantizer):
        self.quantizer = new_quantizer

    def update_cutoff(self, new_cutoff=None, ratio=None):
        self.quantizer.update_cutoff(new_cutoff=new_cutoff, ratio=ratio)
