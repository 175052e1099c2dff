"""Oracle: causal conv stacks and VQAE wiring (CPU, torch fp32, functional).

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.  Every function works on a
plain ``dict[str, Tensor]`` laid out like the reference's ``state_dict()`` so a
checkpoint of the reference model can be fed straight in.

Follows ``networks/vae.py`` (line numbers cited per function) and
``networks/utils.py:34-42`` (weight norm = ``torch.nn.utils.weight_norm``,
dim 0).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from math import ceil
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
LEAKY_SLOPE = 0.1  # vae.py:99,125,156 -- every activation on the path


# --------------------------------------------------------------------------- #
# weight norm                                                                  #
# --------------------------------------------------------------------------- #
def fold_weight_norm(g: Tensor, v: Tensor) -> Tensor:
    """``w = g * v / ||v||`` with the norm over every dim but 0.

    utils.py:34-42 wraps ``torch.nn.utils.weight_norm`` (default ``dim=0``).
    For ``ConvTranspose1d`` dim 0 is C_in (vae.py:54-56), and that is kept.
    """
    norm = v.reshape(v.shape[0], -1).norm(dim=1).reshape(-1, *([1] * (v.dim() - 1)))
    return v * (g / norm)


def conv_params(sd: Dict[str, Tensor], prefix: str) -> Tuple[Tensor, Optional[Tensor]]:
    """Plain (weight, bias) of the conv stored under ``prefix`` (either
    weight-normed ``weight_g/weight_v`` or a plain ``weight``)."""
    if prefix + "weight_g" in sd:
        w = fold_weight_norm(sd[prefix + "weight_g"], sd[prefix + "weight_v"])
    else:
        w = sd[prefix + "weight"]
    return w, sd.get(prefix + "bias")


# --------------------------------------------------------------------------- #
# primitives                                                                   #
# --------------------------------------------------------------------------- #
def causal_pads(length: int, kernel: int, stride: int, dilation: int) -> Tuple[int, int]:
    """(left, right) zero padding of ``CausalConv1d`` -- vae.py:32 and 39-43.

    The right pad formula uses the *undilated* kernel size exactly as the
    reference does (it only matters when stride > 1).
    """
    left = dilation * (kernel - 1) - stride + 1
    nxt = (length - kernel + left) / stride + 1
    target = (ceil(nxt) - 1) * stride + kernel - left
    return left, target - length


def causal_conv1d(x: Tensor, w: Tensor, b: Optional[Tensor], stride: int = 1,
                  dilation: int = 1, groups: int = 1) -> Tensor:
    """vae.py:34-37."""
    left, right = causal_pads(x.shape[-1], w.shape[-1], stride, dilation)
    return F.conv1d(F.pad(x, (left, right)), w, b, stride=stride,
                    dilation=dilation, groups=groups)


def causal_conv_t1d(x: Tensor, w: Tensor, b: Optional[Tensor], stride: int = 1) -> Tensor:
    """vae.py:58-64: transposed conv, then drop the last ``K - stride`` steps."""
    y = F.conv_transpose1d(x, w, b, stride=stride)
    return y[..., : y.shape[-1] - (w.shape[-1] - stride)]


def upsample_conv1d(x: Tensor, w: Tensor, b: Optional[Tensor], scale: int) -> Tensor:
    """vae.py:86-89: nearest-neighbour upsample, then a ``padding="same"`` conv
    (symmetric, i.e. NOT causal -- kept as the reference has it)."""
    up = x.repeat_interleave(scale, dim=-1)  # == F.interpolate(mode="nearest") for int scale
    return F.conv1d(up, w, b, padding="same")


def leaky(x: Tensor) -> Tensor:
    return F.leaky_relu(x, LEAKY_SLOPE)


def residual_block(x: Tensor, sd: Dict[str, Tensor], prefix: str, dilation: int) -> Tensor:
    """vae.py:113-117 (dropout p=0); the ``depthwise=True`` variant (vae.py:103-105: a per-channel k = 1 conv
    in front of the dilated conv) is recognised by its state-dict keys ``conv1.0.* / conv1.1.*``."""
    w2, b2 = conv_params(sd, prefix + "conv2.conv.")
    if prefix + "conv1.0.conv.weight_v" in sd or prefix + "conv1.0.conv.weight" in sd:
        wd, bd = conv_params(sd, prefix + "conv1.0.conv.")
        w1, b1 = conv_params(sd, prefix + "conv1.1.conv.")
        h = causal_conv1d(causal_conv1d(x, wd, bd, groups=x.shape[1]), w1, b1, dilation=dilation)
    else:
        w1, b1 = conv_params(sd, prefix + "conv1.conv.")
        h = causal_conv1d(x, w1, b1, dilation=dilation)
    return x + causal_conv1d(leaky(h), w2, b2)


# --------------------------------------------------------------------------- #
# model description                                                            #
# --------------------------------------------------------------------------- #
@dataclass
class CodecSpec:
    """Constructor arguments of ``CausalVQAE`` that shape the conv stacks
    (vae.py:205-223) -- defaults are the reference's."""
    in_channels: int = 1
    n_blocks: int = 5
    n_layers_per_block: int = 4
    first_block_channels: int = 32
    codebook_dim: int = 512
    strides: Sequence[int] = (2, 3, 4, 4, 5)
    channel_multiplier: int = 2
    wavelet_decoders: Sequence[bool] = field(default_factory=lambda: [False, True, False, False, False])
    input_format: str = "b l c"
    # BUILD-DEFINED (the reference imports CausalMultiresConv1d at vae.py:7 and never wires it): block i carries a multiresolution
    # layer (wavelets.py:38-96) right behind its resampling conv, whose GELU takes the place of the block's LeakyReLU there
    multires_encoders: Sequence[bool] = False
    multires_decoders: Sequence[bool] = False
    multires_kernel_size: int = 2
    multires_depth: int = 3

    def __post_init__(self):
        if isinstance(self.strides, int):
            self.strides = [self.strides] * self.n_blocks
        assert len(self.strides) == self.n_blocks
        if isinstance(self.wavelet_decoders, bool):
            self.wavelet_decoders = [self.wavelet_decoders] * self.n_blocks
        assert len(self.wavelet_decoders) == self.n_blocks
        for name in ("multires_encoders", "multires_decoders"):
            v = getattr(self, name)
            if isinstance(v, bool):
                setattr(self, name, [v] * self.n_blocks)
            assert len(getattr(self, name)) == self.n_blocks

    @property
    def channel_sizes(self) -> List[int]:
        return [self.first_block_channels * self.channel_multiplier ** i
                for i in range(self.n_blocks + 1)]  # vae.py:253

    @property
    def scale_factor(self) -> int:
        out = 1
        for s in self.strides:
            out *= int(s)
        return out

    @property
    def dilations(self) -> List[int]:
        return [3 ** i for i in range(self.n_layers_per_block - 1)]  # vae.py:128,162

    def decoder_is_wavelet(self, dec_index: int) -> bool:
        """``dec_index`` in 1..n_blocks.  vae.py:240 reverses the user list and
        vae.py:271-272 indexes it ``[i-1]`` with i running n_blocks..1, so the
        user's list is in *decoder order* (SURVEY 5.1)."""
        i = self.n_blocks - dec_index + 1          # the loop variable of vae.py:271
        return bool(list(self.wavelet_decoders)[::-1][i - 1])


# --------------------------------------------------------------------------- #
# stacks                                                                       #
# --------------------------------------------------------------------------- #
def encoder_stages(x: Tensor, sd: Dict[str, Tensor], spec: CodecSpec) -> List[Tensor]:
    """Output of every entry of ``model.encoders`` (vae.py:256-266, 310-311).
    ``x`` is (B, C, L)."""
    outs = []
    w, b = conv_params(sd, "encoders.0.1.conv.")
    x = causal_conv1d(x, w, b)
    outs.append(x)
    for i in range(spec.n_blocks):
        p = f"encoders.{i + 1}.layers."
        for j, d in enumerate(spec.dilations):
            x = leaky(residual_block(x, sd, f"{p}{j}.0.", d))
        w, b = conv_params(sd, f"{p}{len(spec.dilations)}.0.conv.")
        x = causal_conv1d(x, w, b, stride=int(spec.strides[i]))
        if spec.multires_encoders[i]:      # build-defined placement (CodecSpec): multires + its GELU instead of the LeakyReLU
            from . import wavelets as _wv
            q = f"encoders.{i + 1}.multires."
            x = _wv.multires_conv(x, sd[q + "h0"], sd[q + "h1"], sd[q + "w"], spec.multires_depth)
        else:
            x = leaky(x)
        outs.append(x)
    w, b = conv_params(sd, f"encoders.{spec.n_blocks + 1}.conv.")
    x = causal_conv1d(x, w, b)
    outs.append(x)
    return outs


def decoder_stages(z: Tensor, sd: Dict[str, Tensor], spec: CodecSpec) -> List[Tensor]:
    """Output of every entry of ``model.decoders`` (vae.py:269-281, 299-301).
    ``z`` is (B, codebook_dim, T)."""
    from . import wavelets as _wv  # local import: keep module import light

    outs = []
    w, b = conv_params(sd, "decoders.0.conv.")
    x = causal_conv_t1d(z, w, b, stride=1)
    outs.append(x)
    for n in range(1, spec.n_blocks + 1):
        stride = int(spec.strides[spec.n_blocks - n])
        p = f"decoders.{n}."
        if spec.decoder_is_wavelet(n):
            x = _wv.wavelet_layer(x, sd, p + "in_conv.0.", scale_factor=stride)
        else:
            w, b = conv_params(sd, p + "in_conv.0.conv.")
            x = upsample_conv1d(x, w, b, stride)
        if spec.multires_decoders[n - 1]:  # build-defined placement (CodecSpec), decoder order
            x = _wv.multires_conv(x, sd[p + "multires.h0"], sd[p + "multires.h1"], sd[p + "multires.w"], spec.multires_depth)
        else:
            x = leaky(x)
        for j, d in enumerate(spec.dilations):
            x = leaky(residual_block(x, sd, f"{p}layers.{j}.0.", d))
        outs.append(x)
    w, b = conv_params(sd, f"decoders.{spec.n_blocks + 1}.conv.")
    x = causal_conv1d(x, w, b)
    outs.append(x)
    return outs


def encode_latents(x: Tensor, sd: Dict[str, Tensor], spec: CodecSpec) -> Tensor:
    """(B,C,L) waveform -> (B,T,D) pre-quantiser latents (vae.py:308-313)."""
    if spec.input_format == "b l c":
        x = x.transpose(1, 2)
    return encoder_stages(x, sd, spec)[-1].transpose(1, 2).contiguous()


def decode_latents(zq: Tensor, sd: Dict[str, Tensor], spec: CodecSpec) -> Tensor:
    """(B,T,D) quantised latents -> waveform in the model's input format
    (vae.py:320-321, 299-303)."""
    y = decoder_stages(zq.transpose(1, 2).contiguous(), sd, spec)[-1]
    if spec.input_format == "b l c":
        y = y.transpose(1, 2)
    return y


def vqae_forward(x: Tensor, sd: Dict[str, Tensor], spec: CodecSpec, codebooks: Tensor,
                 codebook_n: Optional[int] = None):
    """Whole ``CausalVQAE.forward`` in eval mode (vae.py:293-305): returns
    ``(y, commit_loss, index)`` with ``index`` int64 (B,T,Q)."""
    from . import rvq as _rvq

    z = encode_latents(x, sd, spec)
    zq, index, commit = _rvq.residual_quantize(z, codebooks, codebook_n)
    return decode_latents(zq, sd, spec), commit, index


# --------------------------------------------------------------------------- #
# random-init parameters in the reference's state-dict layout                  #
# --------------------------------------------------------------------------- #
def init_state_dict(spec: CodecSpec, seed: int = 0) -> Dict[str, Tensor]:
    """Seeded random parameters with the reference's key names and shapes
    (SURVEY 8b "State-dict").  Values follow torch's default conv init
    (kaiming-uniform with a=sqrt(5) => U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for
    both weight and bias) and weight-norm's ``g = ||v||`` so the folded weight
    equals ``v`` at init -- the same distribution family the reference starts
    from, though not the same random stream."""
    gen = torch.Generator().manual_seed(seed)
    sd: Dict[str, Tensor] = {}

    def add_conv(prefix: str, c_out: int, c_in: int, k: int, transposed: bool = False,
                 weight_norm: bool = True):
        shape = (c_in, c_out, k) if transposed else (c_out, c_in, k)
        fan_in = shape[1] * k
        bound = 1.0 / (fan_in ** 0.5)
        v = (torch.rand(shape, generator=gen) * 2 - 1) * bound
        bias = (torch.rand(c_out, generator=gen) * 2 - 1) * bound
        if weight_norm:
            sd[prefix + "bias"] = bias
            sd[prefix + "weight_g"] = v.reshape(shape[0], -1).norm(dim=1).reshape(-1, 1, 1)
            sd[prefix + "weight_v"] = v
        else:
            sd[prefix + "weight"] = v
            sd[prefix + "bias"] = bias

    def add_multires(prefix: str, channels: int):
        # CausalMultiresConv1d.__init__ (wavelets.py:60-77): uniform(-1, 1) scaled by sqrt(2) / (2 k) resp. sqrt(2 / (2 depth + 4))
        k, depth = spec.multires_kernel_size, spec.multires_depth
        sd[prefix + "h0"] = (torch.rand(channels, 1, k, generator=gen) * 2 - 1) * (2.0 ** 0.5 / (2 * k))
        sd[prefix + "h1"] = (torch.rand(channels, 1, k, generator=gen) * 2 - 1) * (2.0 ** 0.5 / (2 * k))
        sd[prefix + "w"] = (torch.rand(channels, depth + 2, generator=gen) * 2 - 1) * (2.0 / (2 * depth + 4)) ** 0.5

    ch = spec.channel_sizes
    nd = len(spec.dilations)
    add_conv("encoders.0.1.conv.", ch[0], spec.in_channels, 7)
    for i in range(spec.n_blocks):
        p = f"encoders.{i + 1}.layers."
        for j in range(nd):
            add_conv(f"{p}{j}.0.conv1.conv.", ch[i], ch[i], 7)
            add_conv(f"{p}{j}.0.conv2.conv.", ch[i], ch[i], 1)
        add_conv(f"{p}{nd}.0.conv.", ch[i + 1], ch[i], 2 * int(spec.strides[i]) + 1)
        if spec.multires_encoders[i]:
            add_multires(f"encoders.{i + 1}.multires.", ch[i + 1])
    add_conv(f"encoders.{spec.n_blocks + 1}.conv.", spec.codebook_dim, ch[-1], 3)

    add_conv("decoders.0.conv.", ch[-1], spec.codebook_dim, 7, transposed=True)
    for n in range(1, spec.n_blocks + 1):
        i = spec.n_blocks - n + 1
        stride = int(spec.strides[i - 1])
        p = f"decoders.{n}."
        if spec.decoder_is_wavelet(n):
            hidden = 4 * ch[i - 1]
            n_points = 2 * stride * 4
            add_conv(p + "in_conv.0.conv_in.", hidden, ch[i], 2 * stride + 1, weight_norm=False)
            add_conv(p + "in_conv.0.conv_out.", ch[i - 1], hidden, 3, weight_norm=False)
            space = torch.linspace(-10, 10, n_points).reshape(1, 1, 1, n_points)
            sd[p + "in_conv.0.space"] = space
            sd[p + "in_conv.0.cos_kernel"] = torch.cos(space)
            sd[p + "in_conv.0.wavelet_scale"] = torch.full((1, hidden, 1, 1), 40.0)
        else:
            add_conv(p + "in_conv.0.conv.", ch[i - 1], ch[i], 2 * stride + 1)
        if spec.multires_decoders[n - 1]:
            add_multires(p + "multires.", ch[i - 1])
        for j in range(nd):
            add_conv(f"{p}layers.{j}.0.conv1.conv.", ch[i - 1], ch[i - 1], 7)
            add_conv(f"{p}layers.{j}.0.conv2.conv.", ch[i - 1], ch[i - 1], 1)
    add_conv(f"decoders.{spec.n_blocks + 1}.conv.", spec.in_channels, ch[0], 7)
    return sd
