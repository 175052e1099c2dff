"""Drop-in discriminators executed by libagx.

Mirrors ``networks/discriminator.py``: ``WaveformDiscriminatorBlock`` (:7-57),
``WaveFormDiscriminator`` (:59-84), ``STFTDiscriminatorBlock`` (:87-117), ``STFTDiscriminator``
(:119-202) and ``discriminator_generator_loss`` (:204-246) -- same constructor arguments, attributes
(``name``, ``layers`` / ``blocks``), return values ``(outputs, features)`` and ``state_dict`` keys
(old-style ``torch.nn.utils.spectral_norm``: ``bias, weight_orig, weight_u, weight_v``).

Forward, all on HIP kernels through the C ABI:

* spectral norm  -> ``agx_spectral_sigma`` (one power iteration in training mode, buffers updated in
  place) + ``agx_conv_pack_sigma`` / ``agx_conv2d_pack`` (1 / sigma folded into the packed image);
* waveform block -> ``agx_avgpool1d`` + grouped unpadded convs (``AGX_CONV_PADDED``), LeakyReLU fused;
* STFT block     -> ``agx_stft_forward`` (framed DFT as a polyphase conv on the MFMA kernel) +
  ``agx_conv2d_forward`` (kernel rows folded into virtual channels of the 1-D MFMA conv);
* loss           -> ``agx_reduce_mean`` / ``agx_reduce_mean_backward`` (every mean of the hinge and
  feature-matching terms, with hand-written gradients).

Backward through the discriminator bodies runs on the HIP kernels as well (``_STFTDiscNative``,
``_WaveBlockNative``); an activation other than LeakyReLU has no backward kernel and raises -- there is no
ATen fallback.
"""
from __future__ import annotations

import warnings
from typing import List, Optional, Sequence, Tuple

import torch
from torch import nn

from . import ops
from ._lib import CONV_PADDED, EPI_LEAKY_PRE, IMPL_AUTO, IMPL_MFMA_BF16X3, AgxError, needs_grad
from .quantizer import tuple_checker

Tensor = torch.Tensor


def _slope(act: nn.Module) -> float:
    if not isinstance(act, nn.LeakyReLU):
        raise NotImplementedError("only LeakyReLU is fused into the discriminator convs")
    return float(act.negative_slope)


class _SNConv(nn.Module):
    """Parameter holder standing where the reference has ``spectral_norm(Conv1d / Conv2d)`` (or the plain
    conv when ``norm`` is not "spectral").  Initialised by building that very torch module, so the same
    seed gives the same parameters and ``u`` / ``v`` vectors as the reference."""

    def __init__(self, conv: nn.Module, norm: str = "spectral"):
        super().__init__()
        self.norm = norm
        self.eps = 1e-12
        self.nd = 2 if isinstance(conv, nn.Conv2d) else 1
        self.in_channels, self.out_channels = conv.in_channels, conv.out_channels
        self.kernel_size, self.stride, self.padding, self.groups = conv.kernel_size, conv.stride, conv.padding, conv.groups
        if norm == "spectral":
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                conv = nn.utils.spectral_norm(conv)
            self.bias = conv.bias
            self.weight_orig = conv.weight_orig
            self.register_buffer("weight_u", conv.weight_u.detach().clone())
            self.register_buffer("weight_v", conv.weight_v.detach().clone())
        elif norm == "weight":      # add_util_norm(conv, "weight") = old-style torch.nn.utils.weight_norm (utils.py:34-42)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                conv = nn.utils.weight_norm(conv)
            self.bias, self.weight_g, self.weight_v = conv.bias, conv.weight_g, conv.weight_v
        else:
            self.bias, self.weight = conv.bias, conv.weight
        self._key, self._packed, self._iter = None, None, 0
        self.impl = IMPL_AUTO      # IMPL_MFMA_BF16X3: Conv2d layers with a bf16x3 form run on it (set_arithmetic)
        self.ring_only = False     # ... only where the library has the bf16x3 RING kernel for the layer AND the feature map
        self._impl_of = {}         #     (h, w) -> the arithmetic this layer runs at that map size
        self._tape = None          # (sigma, u, v) of the latest forward (native backward)

    @property
    def raw_weight(self) -> Tensor:
        """The tensor the kernels pack: weight_orig (spectral: 1 / sigma is folded in by the pack kernel), the plain
        weight, or -- weight norm, a non-default option -- g v / |v| formed by a few ATen ops on the parameters."""
        if self.norm == "spectral":
            return self.weight_orig
        if self.norm == "weight":
            key = (self.weight_g._version, self.weight_v._version, self.weight_g.data_ptr(), self.weight_v.data_ptr())
            if getattr(self, "_wn_key", None) != key:
                with torch.no_grad():
                    self._wn_w = torch._weight_norm(self.weight_v, self.weight_g, 0).contiguous()
                self._wn_key = key
            return self._wn_w
        return self.weight

    def _finish_grads(self, gl):
        """[dbias?, dW] -> the gradients of grad_params(): the weight-norm chain rule on the (small) weight tensors,
        dg = <dW, v> / |v|, dv = (g / |v|) (dW - v <dW, v> / |v|^2) per output channel."""
        if self.norm != "weight":
            return gl
        dw = gl[-1]
        v, g = self.weight_v.detach(), self.weight_g.detach()
        dims = tuple(range(1, v.dim()))
        nrm = v.norm(2, dim=dims, keepdim=True)
        dot = (dw * v).sum(dim=dims, keepdim=True)
        return list(gl[:-1]) + [dot / nrm, (g / nrm) * (dw - v * dot / (nrm * nrm))]

    def _sigma(self) -> Optional[Tensor]:
        if self.norm != "spectral":
            return None
        n_iter = 1 if self.training else 0
        self._iter += n_iter
        return ops.spectral_sigma(self.weight_orig.detach(), self.weight_u, self.weight_v, n_iter, self.eps)

    def impl_for(self, b: int, h: int, w: int, bwd: bool = False) -> int:
        """The arithmetic of this Conv2d layer's forward (or backward-data: ``bwd``) on an (h, w) input map: ``self.impl``, except
        that in ring-only mode an op / map the bf16x3 ring kernel does not cover (or covers badly: narrow maps) stays on its fp32
        kernel -- forward and backward-data pack their own images, so the two directions choose independently."""
        if self.impl != IMPL_MFMA_BF16X3 or not self.ring_only:
            return self.impl
        key = (h, w, bwd)
        if key not in self._impl_of:
            d = ops.conv2d_desc(b, self.in_channels, self.out_channels, h, w, self.kernel_size[0], self.kernel_size[1],
                                self.stride, self.padding, 0, 0.0, IMPL_MFMA_BF16X3)
            name = ops.conv2d_bwd_data_kernel_name(d) if bwd else ops.conv2d_kernel_name(d)
            self._impl_of[key] = IMPL_MFMA_BF16X3 if name.startswith("conv2d_b3") else IMPL_AUTO
        return self._impl_of[key]

    def packed(self, make_desc, pack_plain, pack_sigma, impl=None) -> Tensor:
        """Packed image; rebuilt when the weight, the buffers or the mode changed (always in training mode:
        the power iteration moves sigma)."""
        w = self.raw_weight
        impl = self.impl if impl is None else impl
        key = (w.data_ptr(), w._version, self.training, self._iter, impl,
               getattr(self, "_wn_key", None) if self.norm != "spectral" else (self.weight_u._version, self.weight_v._version))
        if self.training or key != self._key:
            sigma = self._sigma()
            # what this forward normalised with (the native backward needs exactly these)
            self._tape = None if sigma is None else (sigma, self.weight_u.clone(), self.weight_v.clone())
            desc = make_desc()
            self._packed = pack_plain(desc, w.detach()) if sigma is None else pack_sigma(desc, w.detach(), sigma)
            self._key = (w.data_ptr(), w._version, self.training, self._iter, impl,
                         getattr(self, "_wn_key", None) if self.norm != "spectral" else (self.weight_u._version, self.weight_v._version))
        return self._packed

    # -- 1-D --------------------------------------------------------------------------------------
    def run1d(self, x: Tensor, slope: Optional[float]) -> Tensor:
        b, _, length = x.shape

        def desc(batch=b, l_in=length):
            return ops.conv_desc(CONV_PADDED, batch, self.in_channels, self.out_channels, l_in, self.kernel_size[0],
                                 self.stride[0], 1, EPI_LEAKY_PRE if slope is not None else 0, slope or 0.0,
                                 groups=self.groups, padding=self.padding[0])

        packed = self.packed(lambda: desc(1, 1 << 20), ops.conv_pack, ops.conv_pack_sigma)
        return ops.conv_forward(desc(), x, packed, None if self.bias is None else self.bias.detach())

    # -- 2-D --------------------------------------------------------------------------------------
    def run2d(self, x: Tensor, slope: Optional[float]) -> Tensor:
        b, _, h, w = x.shape
        impl = self.impl_for(b, h, w)

        def desc(batch=b, hh=h, ww=w):
            return ops.conv2d_desc(batch, self.in_channels, self.out_channels, hh, ww, self.kernel_size[0],
                                   self.kernel_size[1], self.stride, self.padding,
                                   EPI_LEAKY_PRE if slope is not None else 0, slope or 0.0, impl)

        packed = self.packed(lambda: desc(1, 64, 64), ops.conv2d_pack, ops.conv2d_pack, impl)
        return ops.conv2d_forward(desc(), x, packed, None if self.bias is None else self.bias.detach())


    def desc2d(self, x: Tensor, slope: Optional[float] = None, bwd: bool = False):
        b, _, h, w = x.shape
        return ops.conv2d_desc(b, self.in_channels, self.out_channels, h, w, self.kernel_size[0], self.kernel_size[1],
                               self.stride, self.padding, EPI_LEAKY_PRE if slope is not None else 0, slope or 0.0,
                               self.impl_for(b, h, w, bwd))

    def bwd2d(self, x: Tensor, dy: Tensor, tape, need_dx: bool = True, add: Optional[Tensor] = None,
              mask: Optional[Tensor] = None, slope: float = 0.2):
        """(dx or None, [dbias, dweight]) of this layer for the forward that produced ``tape``."""
        desc = self.desc2d(x, bwd=True)
        w = self.raw_weight.detach()
        sigma, u, v = tape if tape is not None else (None, None, None)
        dw, db = ops.conv2d_bwd_weight(desc, x, dy, w, sigma, u, v, want_bias=self.bias is not None)
        dx = None
        if need_dx:
            few = (self.in_channels * self.stride[0] * self.stride[1] < 32 and self.stride == (1, 1) and mask is None
                   and self.out_channels % 16 == 0 and self.in_channels * self.kernel_size[1] >= 8
                   and self.kernel_size[0] - 1 - self.padding[0] >= 0)
            if few:    # the 2-channel first conv: kw * Cin rows on the MFMA tiles instead of Cin on the direct kernel
                dx = ops.conv2d_bwd_data_fewchannels(desc, dy, w, sigma, add)
            else:
                dx = ops.conv2d_bwd_data(desc, dy, ops.conv2d_pack_bwd(desc, w, sigma), mask, slope, add)
        return dx, self._finish_grads([db, dw] if self.bias is not None else [dw])

    def desc1d(self, x: Tensor, slope: Optional[float] = None):
        return ops.conv_desc(CONV_PADDED, x.shape[0], self.in_channels, self.out_channels, x.shape[2],
                             self.kernel_size[0], self.stride[0], 1, EPI_LEAKY_PRE if slope is not None else 0,
                             slope or 0.0, groups=self.groups, padding=self.padding[0])

    def bwd1d(self, x: Tensor, dz: Tensor, tape, need_dx: bool = True, add: Optional[Tensor] = None,
              mask: Optional[Tensor] = None, slope: float = 0.2):
        """(dx or None, [dbias, dweight]) of this 1-D layer for the forward that produced ``tape``."""
        desc = self.desc1d(x)
        w = self.raw_weight.detach()
        sigma, u, v = tape if tape is not None else (None, None, None)
        if self.groups > 1:
            dw, db = ops.conv_grouped_bwd_weight(desc, x, dz, want_bias=self.bias is not None)
            dx = ops.conv_grouped_bwd_data(desc, dz, w, sigma, add, mask, slope) if need_dx else None
        else:
            dw, _, db = ops.conv_bwd_weight(desc, x, dz, w, None, want_bias=self.bias is not None)
            dx = None
            if need_dx:
                pk = ops.conv_pack_bwd(desc, w) if sigma is None else ops.conv_pack_bwd_sigma(desc, w, sigma)
                dx = ops.conv_bwd_data(desc, dz, pk, add, mask, slope)
        if sigma is not None:
            ops.spectral_grad_(dw, w, sigma, u, v)
        return dx, self._finish_grads([db, dw] if self.bias is not None else [dw])

    def grad_params(self):
        weights = [self.weight_g, self.weight_v] if self.norm == "weight" else [self.raw_weight]
        return ([self.bias] if self.bias is not None else []) + weights


def set_arithmetic(module: nn.Module, mode: str = "fp32") -> nn.Module:
    """Arithmetic of the STFT discriminators' Conv2d layers (forward, backward-data and the weight-gradient contraction):
    ``"bf16x3_ring"``: the 3 x 3 stride-1 "same" layers with Cin % 32 == 0 and 32 / 64 / a multiple of 128 output channels run on
    the bf16x3 ring kernel (csrc/conv_b3.hip: conv2d_b3_kernel, DESIGN 4.12: fp32-class accuracy, 1.3-1.5 x the fp32 ring) on
    feature maps wide enough for its tiles; every other layer / map keeps its fp32 kernel; ``"bf16x3"``: every Conv2d layer with a bf16x3 form (Cin % 16 == 0, Cout >= 32) -- the ring
    where it exists, the round-1 kernels elsewhere; default ``"fp32"``."""
    if mode not in ("fp32", "bf16x3", "bf16x3_ring"):
        raise ValueError(f"unknown arithmetic {mode!r}")
    for m in module.modules():
        if isinstance(m, _SNConv):
            on = mode != "fp32" and m.nd == 2
            m.impl = IMPL_MFMA_BF16X3 if on else IMPL_AUTO
            m.ring_only = mode == "bf16x3_ring"      # decided per feature-map size by the library (impl_for)
            m._impl_of = {}
    return module


class WaveformDiscriminatorBlock(nn.Module):
    """discriminator.py:7-57."""

    def __init__(self, in_channels, channel_sizes=(16, 64, 256, 512, 1024, 1024, 1024),
                 kernel_sizes=(15, 41, 41, 41, 41, 5, 3), strides=(1, 4, 4, 4, 4, 1, 1),
                 groups=(1, 4, 16, 64, 256, 1, 1), activation=None, scale=1, norm="spectral", apply_sigmoid=True):
        super().__init__()
        activation = nn.LeakyReLU(0.2) if activation is None else activation
        n_steps = len(channel_sizes)
        self.channel_sizes = [in_channels] + list(channel_sizes)
        self.kernel_sizes = tuple_checker(kernel_sizes, n_steps)
        self.strides = tuple_checker(strides, n_steps)
        self.groups = tuple_checker(groups, n_steps)
        self.scale = scale
        layers: List[nn.Module] = [nn.AvgPool1d(2 * scale, stride=scale, padding=scale)]
        for i in range(n_steps - 1):
            conv = nn.Conv1d(self.channel_sizes[i], self.channel_sizes[i + 1], self.kernel_sizes[i],
                             stride=self.strides[i], groups=self.groups[i])
            layers.append(nn.Sequential(_SNConv(conv, norm), activation))
        layers.append(_SNConv(nn.Conv1d(channel_sizes[-1], 1, self.kernel_sizes[-1], stride=self.strides[-1],
                                        groups=self.groups[-1]), norm))
        self.layers = nn.ModuleList(layers)
        self.final_activation = nn.Sigmoid() if apply_sigmoid else nn.Identity()

    def _hip(self, x: Tensor) -> List[Tensor]:
        pool = self.layers[0]
        x = ops.avgpool1d(x, pool.kernel_size[0] if isinstance(pool.kernel_size, tuple) else pool.kernel_size,
                          pool.stride[0] if isinstance(pool.stride, tuple) else pool.stride,
                          pool.padding[0] if isinstance(pool.padding, tuple) else pool.padding)
        feats = [x]
        for layer in list(self.layers)[1:]:
            if isinstance(layer, nn.Sequential):
                x = layer[0].run1d(x, _slope(layer[1]))
            else:
                x = layer.run1d(x, None)
            feats.append(x)
        out = ops.sigmoid(x) if isinstance(self.final_activation, nn.Sigmoid) else x.clone()
        return [out] + feats

    def forward(self, x: Tensor):
        if needs_grad(x, self):
            if not all(isinstance(l[1], nn.LeakyReLU) for l in self.layers if isinstance(l, nn.Sequential)):
                raise AgxError("WaveformDiscriminatorBlock: the backward kernels fuse the LeakyReLU gradient only; another "
                               "activation can run forward (torch.no_grad) but has no backward -- there is no ATen fallback")
            flat = _WaveBlockNative.apply(self, x, *list(self.parameters()))
            return flat[0], list(flat[1:])
        flat = self._hip(x)
        return flat[0], list(flat[1:])


class WaveFormDiscriminator(nn.Module):
    """discriminator.py:59-84."""

    def __init__(self, in_channels, name="waveform_discriminator", n_blocks=3, scalefactor_per_block=2,
                 norm="spectral"):
        super().__init__()
        self.name = name
        scales = [scalefactor_per_block ** i for i in range(n_blocks)]
        self.layers = nn.ModuleList([WaveformDiscriminatorBlock(in_channels, scale=s, norm=norm) for s in scales])

    def forward(self, x: Tensor):
        features, outputs = [], []
        for layer in self.layers:
            out, layer_features = layer(x)
            outputs.append(out)
            features.extend(layer_features)
        return outputs, features


class STFTDiscriminatorBlock(nn.Module):
    """discriminator.py:87-117 (conv 3x3 -> activation -> strided conv; the residual is commented out there)."""

    def __init__(self, in_channels, channel_multiplier, stride, kernel_size=None, padding=None, activation=None,
                 norm="spectral"):
        super().__init__()
        activation = nn.LeakyReLU(0.2) if activation is None else activation
        if kernel_size is None:
            kernel_size = (stride[0] + 2, stride[1] + 2)
        if padding is None:
            padding = ((kernel_size[0] - 1) // 2, (kernel_size[1] - 1) // 2)
        self.layers = nn.Sequential(
            _SNConv(nn.Conv2d(in_channels, in_channels, kernel_size=3, padding=1), norm),
            activation,
            _SNConv(nn.Conv2d(in_channels, in_channels * channel_multiplier, stride=stride, kernel_size=kernel_size,
                              padding=padding), norm))

    def forward(self, x: Tensor) -> Tensor:
        x = self.layers[0].run2d(x, _slope(self.layers[1]))
        return self.layers[2].run2d(x, None)


class STFTDiscriminator(nn.Module):
    """discriminator.py:119-202."""

    def __init__(self, in_channels=2, first_channel_size=32, channel_multipliers=(2, 2, 1, 2, 1, 2),
                 strides=((1, 2), (2, 2)) * 3, win_length=1024, n_fft=None, hop_length=None, feature_multiplier=1,
                 normalize_stft=True, norm="spectral", base_name="stft_discriminator", apply_sigmoid=True):
        super().__init__()
        self.win_length = win_length
        self.n_fft = win_length if n_fft is None else n_fft
        self.hop_length = win_length // 4 if hop_length is None else hop_length
        if self.n_fft != win_length or self.hop_length * 4 != win_length:
            raise NotImplementedError("the HIP STFT covers the reference wiring: n_fft = win_length, hop = win_length / 4")
        self.normalize_stft = normalize_stft
        self.feature_multiplier = feature_multiplier
        self.name = f"{base_name}_{win_length}"
        self.num_blocks = len(channel_multipliers)
        self.first_conv = _SNConv(nn.Conv2d(in_channels, first_channel_size, kernel_size=7, padding=3), norm)
        blocks, ch = [], first_channel_size
        for mult, stride in zip(channel_multipliers, strides):
            blocks.append(STFTDiscriminatorBlock(ch, mult, tuple(stride), norm=norm))
            ch *= mult
        self.blocks = nn.ModuleList(blocks)
        fk = win_length // (2 ** (self.num_blocks + 1))
        self.final_conv = _SNConv(nn.Conv2d(ch, 1, kernel_size=(1, fk), padding=(0, (fk - 1) // 2)), norm)
        self.final_activation = nn.Sigmoid() if apply_sigmoid else nn.Identity()

    def _hip(self, x: Tensor) -> List[Tensor]:
        x = ops.stft(x.squeeze(1), self.n_fft, self.normalize_stft)          # (B, 2, T, F)
        x = self.first_conv.run2d(x, None)
        feats = [x]
        for block in self.blocks:
            x = block(x)
            feats.append(x)
        x = self.final_conv.run2d(x, None)
        out = ops.sigmoid(x) if isinstance(self.final_activation, nn.Sigmoid) else x.clone()
        return [out] + feats

    def forward(self, x: Tensor):
        if needs_grad(x, self):
            if not all(isinstance(b.layers[1], nn.LeakyReLU) for b in self.blocks):
                raise AgxError("STFTDiscriminator: the backward kernels fuse the LeakyReLU gradient only; another activation "
                               "can run forward (torch.no_grad) but has no backward -- there is no ATen fallback")
            flat = _STFTDiscNative.apply(self, x, *list(self.parameters()))
            return [flat[0]], list(flat[1:])
        flat = self._hip(x)
        return [flat[0]], list(flat[1:])


class _WaveBlockNative(torch.autograd.Function):
    """WaveformDiscriminatorBlock forward + hand-written backward on the HIP kernels (same scheme as
    _STFTDiscNative: per layer one dW call and one bwd-data call with the feature gradient and the LeakyReLU
    gradient of the layer below fused in)."""

    @staticmethod
    def forward(ctx, blk, x: Tensor, *params: Tensor):
        pool = blk.layers[0]
        one = lambda t: t[0] if isinstance(t, tuple) else t  # noqa: E731
        ctx.pool = (one(pool.kernel_size), one(pool.stride), one(pool.padding))
        with torch.no_grad():
            h = ops.avgpool1d(x.detach(), *ctx.pool)
            feats, convs, slopes, tapes = [h], [], [], []
            for layer in list(blk.layers)[1:]:
                conv, act = (layer[0], layer[1]) if isinstance(layer, nn.Sequential) else (layer, None)
                sl = None if act is None else _slope(act)
                h = conv.run1d(h, sl)
                tapes.append(conv._tape)
                convs.append(conv)
                slopes.append(sl)
                feats.append(h)
            sig = isinstance(blk.final_activation, nn.Sigmoid)
            out = ops.sigmoid(h) if sig else h.clone()
        ctx.convs, ctx.slopes, ctx.tapes, ctx.sig, ctx.params, ctx.l_in = convs, slopes, tapes, sig, params, x.shape[-1]
        ctx.save_for_backward(out, *feats)
        return (out, *feats)

    @staticmethod
    def backward(ctx, g_out: Optional[Tensor], *g_feats: Optional[Tensor]):
        saved = ctx.saved_tensors
        out, feats = saved[0], saved[1:]
        convs, slopes, tapes = ctx.convs, ctx.slopes, ctx.tapes
        gf = [None if g is None else g.contiguous() for g in g_feats]
        grads = {}
        # gradient w.r.t. the last conv's output (a feature too): from the sigmoid output and from the feature loss
        dz = None
        if g_out is not None:
            dz = ops.sigmoid_backward(g_out.contiguous(), out) if ctx.sig else g_out.contiguous()
        if gf[-1] is not None:
            dz = gf[-1] if dz is None else dz + gf[-1]
        for i in range(len(convs) - 1, -1, -1):
            x_in = feats[i]                      # input of conv i (= feature i: pooled input or the layer below)
            if dz is None:
                dz = gf[i]                       # nothing from above: the feature gradient, if any, starts the chain
                if dz is not None and i > 0 and slopes[i - 1] is not None:
                    dz = torch.where(x_in > 0, dz, dz * slopes[i - 1])
                continue
            below = slopes[i - 1] if i > 0 else None      # activation that produced x_in
            need_dx = i > 0 or ctx.needs_input_grad[1]
            dxl, gl = convs[i].bwd1d(x_in, dz, tapes[i], need_dx=need_dx, add=gf[i],
                                     mask=x_in if below is not None else None, slope=below or 0.0)
            for p_, g_ in zip(convs[i].grad_params(), gl):
                grads[p_] = g_
            dz = dxl
        dx = None
        if ctx.needs_input_grad[1] and dz is not None:
            dx = ops.avgpool1d_backward(dz, ctx.l_in, *ctx.pool)
        return (None, dx, *[grads.get(p_) for p_ in ctx.params])


class _STFTDiscNative(torch.autograd.Function):
    """STFTDiscriminator forward + hand-written backward, everything on the HIP kernels: per layer one
    weight-gradient call (spectral-norm chain rule included) and one backward-data call whose epilogue adds
    the gradient arriving at that feature map from the feature-matching loss and applies the LeakyReLU
    gradient of the layer below; the STFT adjoint closes the path to the waveform."""

    @staticmethod
    def forward(ctx, disc, x: Tensor, *params: Tensor):
        with torch.no_grad():
            xd = x.detach()
            spec = ops.stft(xd.squeeze(1), disc.n_fft, disc.normalize_stft)
            convs, tapes, acts = [disc.first_conv], [], []
            h = disc.first_conv.run2d(spec, None)
            tapes.append(disc.first_conv._tape)
            feats = [h]
            for blk in disc.blocks:
                c0, act, c2 = blk.layers[0], blk.layers[1], blk.layers[2]
                a = c0.run2d(h, _slope(act))
                tapes.append(c0._tape)
                h = c2.run2d(a, None)
                tapes.append(c2._tape)
                convs += [c0, c2]
                acts.append(a)
                feats.append(h)
            z = disc.final_conv.run2d(h, None)
            tapes.append(disc.final_conv._tape)
            convs.append(disc.final_conv)
            sig = isinstance(disc.final_activation, nn.Sigmoid)
            out = ops.sigmoid(z) if sig else z.clone()
        ctx.disc, ctx.convs, ctx.tapes, ctx.sig, ctx.params = disc, convs, tapes, sig, params
        ctx.length = x.shape[-1]
        ctx.n_blocks = len(acts)
        ctx.save_for_backward(spec, out, *feats, *acts)
        return (out, *feats)

    @staticmethod
    def backward(ctx, g_out: Optional[Tensor], *g_feats: Optional[Tensor]):
        disc, convs, tapes = ctx.disc, ctx.convs, ctx.tapes
        saved = ctx.saved_tensors
        nb = ctx.n_blocks
        spec, out = saved[0], saved[1]
        feats, acts = saved[2:3 + nb], saved[3 + nb:]
        grads = {}

        def put(conv, gl):
            for p_, g_ in zip(conv.grad_params(), gl):
                grads[p_] = g_

        gf = [None if g is None else g.contiguous() for g in g_feats]
        dh = None
        if g_out is not None:
            dz = ops.sigmoid_backward(g_out.contiguous(), out) if ctx.sig else g_out.contiguous()
            dh, gl = convs[-1].bwd2d(feats[-1], dz, tapes[-1], add=gf[-1])
            put(convs[-1], gl)
        else:
            dh = gf[-1]
        for i in range(nb - 1, -1, -1):
            c0, c2 = convs[1 + 2 * i], convs[2 + 2 * i]
            slope = _slope(disc.blocks[i].layers[1])
            if dh is None:                               # nothing arrives at this feature map or above
                dh = gf[i]
                continue
            da, gl = c2.bwd2d(acts[i], dh, tapes[2 + 2 * i], mask=acts[i], slope=slope)
            put(c2, gl)
            dh, gl = c0.bwd2d(feats[i], da, tapes[1 + 2 * i], add=gf[i])
            put(c0, gl)
        dx = None
        if dh is not None:
            need_dx = ctx.needs_input_grad[1]
            dspec, gl = convs[0].bwd2d(spec, dh, tapes[0], need_dx=need_dx)
            put(convs[0], gl)
            if need_dx:
                dx = ops.stft_backward(dspec, ctx.length, disc.n_fft, disc.normalize_stft).unsqueeze(1)
        return (None, dx, *[grads.get(p_) for p_ in ctx.params])


class _Mean(torch.autograd.Function):
    """One mean of the loss on ``agx_reduce_mean`` with its hand-written gradient."""

    @staticmethod
    def forward(ctx, mode: int, x: Tensor, y: Optional[Tensor]):
        ctx.mode = mode
        ctx.save_for_backward(x, y) if y is not None else ctx.save_for_backward(x)
        return ops.reduce_mean(x.detach(), mode, None if y is None else y.detach())

    @staticmethod
    def backward(ctx, g: Tensor):
        saved = ctx.saved_tensors
        x, y = saved[0], (saved[1] if len(saved) > 1 else None)
        want_dy = y is not None and ctx.needs_input_grad[2]
        dx, dy = ops.reduce_mean_backward(x, ctx.mode, g.contiguous(), y, want_dy)
        return None, (dx if ctx.needs_input_grad[1] else None), dy


def _mean(mode: int, x: Tensor, y: Optional[Tensor] = None) -> Tensor:
    return _Mean.apply(mode, x, y)


class _FeatureMeans(torch.autograd.Function):
    """Both means of one feature-matching term -- mean|x - y| and the scale mean|x + 1e-3| -- in one pass over the
    pair of feature maps, and their gradients in one pass (``agx_feature_means``): the values and gradients of
    ``_mean(REDUCE_L1, x, y)`` and ``_mean(REDUCE_ABS_EPS, x)`` bit for bit, at half the memory traffic."""

    @staticmethod
    def forward(ctx, x: Tensor, y: Tensor):
        ctx.save_for_backward(x, y)
        return ops.feature_means(x.detach(), y.detach())

    @staticmethod
    def backward(ctx, g: Tensor):
        x, y = ctx.saved_tensors
        dx, dy = ops.feature_means_backward(x, y, g.contiguous(), ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return dx, dy


def discriminator_generator_loss(original: Tensor, reconstruction: Tensor, discriminator: nn.Module,
                                 feature_multipier: float = 100, scale_feature_loss: bool = True):
    """discriminator.py:204-246 (argument spelling kept): three passes through the discriminator, hinge
    GAN terms averaged over its outputs, L1 feature matching."""
    # The reference feeds ``original.clone().requires_grad_()`` and ``reconstruction.detach().clone().requires_grad_()``
    # to the real / detached-fake passes; those clones are local leaves nobody reads, so their input gradients
    # (first-conv backward-data + STFT adjoint, per discriminator) are simply not requested here.  Parameter
    # gradients are unaffected.
    original_d, original_features = discriminator(original.detach())
    reconstruction_d, reconstruction_features = discriminator(reconstruction)
    reconstruction_d2, _ = discriminator(reconstruction.detach())
    k = len(original_d)
    discriminator_loss, generation_loss = 0, 0
    for x, y, y_disc in zip(original_d, reconstruction_d, reconstruction_d2):
        real_d_loss = -_mean(ops.REDUCE_HINGE_REAL, x)
        fake_d_loss = -_mean(ops.REDUCE_HINGE_FAKE, y_disc)
        discriminator_loss = discriminator_loss + (real_d_loss + fake_d_loss) / k
        generation_loss = generation_loss - _mean(ops.REDUCE_MEAN, y) / k
    feature_loss, n_features = 0, len(original_features)
    for x, y in zip(original_features, reconstruction_features):
        if scale_feature_loss:
            pair = _FeatureMeans.apply(x, y)
            feature_loss_i = pair[0] / n_features / pair[1]
        else:
            feature_loss_i = _mean(ops.REDUCE_L1, x, y) / n_features
        feature_loss = feature_loss + feature_loss_i
    generator_loss = generation_loss + feature_multipier * feature_loss
    return generator_loss, discriminator_loss
