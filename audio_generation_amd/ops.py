"""Tensor-level wrappers over the C ABI: allocate outputs with torch, pass raw
device pointers + the current HIP stream to ``libagx``.  PyTorch is plumbing
here (memory, streams); all arithmetic happens in the HIP kernels.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import (CONV_CAUSAL, CONV_SAME, CONV_TRANSPOSED, CONV_UPSAMPLE, EPI_LEAKY_POST,
                   EPI_GELU_PRE, EPI_LEAKY_PRE, EPI_RESIDUAL, IMPL_AUTO, IMPL_DIRECT, IMPL_MFMA, AgxError, ConvDesc)

Tensor = torch.Tensor

# Optional launch observer (bench.py / profiling): an object with
# ``begin(kind, info) -> token`` and ``end(token)`` called around every C-ABI
# compute call.  None (the default) costs one global lookup per call.
_observer = None


def set_observer(obs) -> None:
    global _observer
    _observer = obs


def count_macs(kind: str, macs: int, desc=None) -> None:
    """Report the executed multiply-accumulates of a launch that has no timing hook (backward, 2-D and spectral ops) to the
    observer's optional ``macs(kind, n)`` method -- bench.py's FLOP account of the training step.  A layer whose descriptor
    asks for the bf16x3 arithmetic is reported as ``kind + ":bf16x3"`` (its products run on the bf16 matrix pipe at six
    bf16 flops per fp32-equivalent flop: a different roofline)."""
    if _observer is not None and hasattr(_observer, "macs"):
        if desc is not None and getattr(desc, "impl", 0) == _lib.IMPL_MFMA_BF16X3:
            kind += ":bf16x3"
        _observer.macs(kind, int(macs))


def _conv_macs(desc: "ConvDesc") -> int:
    """Executed (polyphase) MACs of a 1-D conv layer: B * Cin/groups * J * (q * Cout) * Lt."""
    b, cin, cout, lin, k, s = desc.batch, desc.c_in, desc.c_out, desc.l_in, desc.kernel, desc.stride
    g = max(getattr(desc, "groups", 1), 1)
    if desc.kind == CONV_UPSAMPLE:
        pl = (k - 1) // 2
        jmin, jmax = (-pl) // s, (s - 1 + k - 1 - pl) // s
        return b * cin * (jmax - jmin + 1) * s * cout * lin
    if desc.kind == CONV_TRANSPOSED:
        return b * cin * (-(-k // s)) * s * cout * lin
    return b * (cin // g) * cout * k * conv_out_len(desc)


def _conv2d_macs(desc) -> int:
    ho = (desc.h_in + 2 * desc.pad_h - desc.kh) // desc.stride_h + 1
    wo = (desc.w_in + 2 * desc.pad_w - desc.kw) // desc.stride_w + 1
    return desc.batch * desc.c_in * desc.c_out * desc.kh * desc.kw * ho * wo


def conv_kernel_name(desc: "ConvDesc") -> str:
    buf = ctypes.create_string_buffer(96)
    _lib.check(_lib.load().agx_conv_kernel_name(ctypes.byref(desc), buf, len(buf)), "agx_conv_kernel_name")
    return buf.value.decode()


def resblock_kernel_name(desc: "ConvDesc") -> str:
    buf = ctypes.create_string_buffer(96)
    _lib.check(_lib.load().agx_resblock_kernel_name(ctypes.byref(desc), buf, len(buf)), "agx_resblock_kernel_name")
    return buf.value.decode()


def _ptr(t: Optional[Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_gpu(*tensors: Optional[Tensor]) -> None:
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise AgxError("audio_generation_amd runs on the MI355X only: got a tensor on "
                           f"'{t.device}'.  There is no CPU / eager fallback.")
        if t.dtype not in (torch.float32, torch.int64, torch.float64, torch.uint8, torch.bfloat16):   # bfloat16: activation planes
            raise AgxError(f"unsupported dtype {t.dtype}")


def _f32c(t: Tensor) -> Tensor:
    if t.dtype != torch.float32:
        raise AgxError(f"expected float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


# --------------------------------------------------------------------------- conv
def conv_desc(kind: int, batch: int, c_in: int, c_out: int, l_in: int, kernel: int, stride: int = 1,
              dilation: int = 1, epilogue: int = 0, slope: float = 0.1, impl: int = IMPL_AUTO, groups: int = 1,
              padding: int = 0) -> ConvDesc:
    return ConvDesc(kind, batch, c_in, c_out, l_in, kernel, stride, dilation, epilogue, slope, impl, groups, padding)


def conv_out_len(desc: ConvDesc) -> int:
    n = _lib.load().agx_conv_out_len(ctypes.byref(desc))
    if n < 0:
        _lib.check(int(n), "agx_conv_out_len")
    return int(n)


def conv_pack(desc: ConvDesc, v: Tensor, g: Optional[Tensor] = None) -> Tensor:
    """Weight-norm fold + repack (``agx_conv_pack``).  Returns the packed image."""
    lib = _lib.load()
    _need_gpu(v, g)
    n = lib.agx_conv_packed_floats(ctypes.byref(desc))
    if n < 0:
        _lib.check(int(n), "agx_conv_packed_floats")
    v = _f32c(v)
    g = None if g is None else _f32c(g)
    packed = torch.empty(int(n), dtype=torch.float32, device=v.device)
    _lib.check(lib.agx_conv_pack(ctypes.byref(desc), _ptr(v), _ptr(g), _ptr(packed), _stream()),
               "agx_conv_pack")
    return packed


def conv_forward(desc: ConvDesc, x: Tensor, packed: Tensor, bias: Optional[Tensor],
                 res: Optional[Tensor] = None, out: Optional[Tensor] = None) -> Tensor:
    lib = _lib.load()
    _need_gpu(x, packed, bias, res)
    x = _f32c(x)
    if tuple(x.shape) != (desc.batch, desc.c_in, desc.l_in):
        raise AgxError(f"conv_forward: x is {tuple(x.shape)}, descriptor says "
                       f"{(desc.batch, desc.c_in, desc.l_in)}")
    l_out = conv_out_len(desc)
    y = out if out is not None else torch.empty((desc.batch, desc.c_out, l_out), dtype=torch.float32,
                                                 device=x.device)
    if res is not None:
        res = _f32c(res)
        if res.shape != y.shape:
            raise AgxError(f"conv_forward: residual is {tuple(res.shape)}, output is {tuple(y.shape)}")
    if bias is not None:
        bias = _f32c(bias)
    tok = _observer.begin("conv", desc) if _observer is not None else None
    _lib.check(lib.agx_conv_forward(ctypes.byref(desc), _ptr(x), _ptr(packed), _ptr(bias), _ptr(res),
                                    _ptr(y), _stream()), "agx_conv_forward")
    if tok is not None:
        _observer.end(tok)
    return y


# ---------------------------------------------------------------- activation planes (bf16x3, include/agx.h)
def planes_split(x: Tensor) -> Tensor:
    """fp32 (B, C, L) -> activation planes: bf16 (B, C / 8, 3, L, 8), h + m + l == x exactly."""
    lib = _lib.load()
    _need_gpu(x)
    x = _f32c(x)
    b, c, length = x.shape
    if c % 8:
        raise AgxError(f"planes_split: {c} channels (must be a multiple of 8)")
    planes = torch.empty((b, c // 8, 3, length, 8), dtype=torch.bfloat16, device=x.device)
    tok = _observer.begin("other", ("planes_split", 10 * x.numel())) if _observer is not None else None
    _lib.check(lib.agx_planes_split(_ptr(x), _ptr(planes), b, c, length, _stream()), "agx_planes_split")
    if tok is not None:
        _observer.end(tok)
    return planes


def planes_join(planes: Tensor) -> Tensor:
    """The fp32 activation a planes tensor stands for (h + m + l; host-side helper for tests and debugging, torch arithmetic)."""
    b, g, _, length, _ = planes.shape
    p = planes.float()
    return ((p[:, :, 0] + p[:, :, 1]) + p[:, :, 2]).permute(0, 1, 3, 2).reshape(b, g * 8, length).contiguous()


def conv_planes_supported(desc: ConvDesc) -> int:
    """0: no planes path; 1: the layer can read planes; 2: it can also write its output as planes."""
    return int(_lib.load().agx_conv_planes_supported(ctypes.byref(desc)))


def conv_forward_planes(desc: ConvDesc, x_planes: Tensor, packed: Tensor, bias: Optional[Tensor],
                        out_planes: bool = False) -> Tensor:
    """``conv_forward`` of a bf16x3 ring layer whose input is given as activation planes; ``out_planes``: the output is
    returned as planes as well (one-phase layers only)."""
    lib = _lib.load()
    _need_gpu(x_planes, packed, bias)
    if x_planes.dtype != torch.bfloat16 or tuple(x_planes.shape) != (desc.batch, desc.c_in // 8, 3, desc.l_in, 8) \
            or not x_planes.is_contiguous():
        raise AgxError(f"conv_forward_planes: planes are {tuple(x_planes.shape)} {x_planes.dtype}, descriptor says "
                       f"{(desc.batch, desc.c_in // 8, 3, desc.l_in, 8)} bfloat16")
    l_out = conv_out_len(desc)
    bias = None if bias is None else _f32c(bias)
    if out_planes:
        y = torch.empty((desc.batch, desc.c_out // 8, 3, l_out, 8), dtype=torch.bfloat16, device=x_planes.device)
    else:
        y = torch.empty((desc.batch, desc.c_out, l_out), dtype=torch.float32, device=x_planes.device)
    tok = _observer.begin("conv", desc) if _observer is not None else None
    _lib.check(lib.agx_conv_forward_planes(ctypes.byref(desc), _ptr(x_planes), _ptr(packed), _ptr(bias),
                                           None if out_planes else _ptr(y), _ptr(y) if out_planes else None, _stream()),
               "agx_conv_forward_planes")
    if tok is not None:
        _observer.end(tok)
    return y


def conv_pack_bwd(desc: ConvDesc, v: Tensor, g: Optional[Tensor] = None) -> Tensor:
    """Packed image of the layer's backward-data op (``agx_conv_pack_bwd``)."""
    lib = _lib.load()
    _need_gpu(v, g)
    n = lib.agx_conv_bwd_packed_floats(ctypes.byref(desc))
    if n < 0:
        _lib.check(int(n), "agx_conv_bwd_packed_floats")
    v = _f32c(v)
    g = None if g is None else _f32c(g)
    packed = torch.empty(int(n), dtype=torch.float32, device=v.device)
    _lib.check(lib.agx_conv_pack_bwd(ctypes.byref(desc), _ptr(v), _ptr(g), _ptr(packed), _stream()),
               "agx_conv_pack_bwd")
    return packed


def conv_bwd_data(desc: ConvDesc, dy: Tensor, packed_bwd: Tensor, add: Optional[Tensor] = None,
                  mask: Optional[Tensor] = None, slope: float = 0.1) -> Tensor:
    """Gradient w.r.t. the input of the layer ``desc`` describes (forward descriptor)."""
    lib = _lib.load()
    _need_gpu(dy, packed_bwd, add, mask)
    dy = _f32c(dy)
    l_out = conv_out_len(desc)
    if tuple(dy.shape) != (desc.batch, desc.c_out, l_out):
        raise AgxError(f"conv_bwd_data: dy is {tuple(dy.shape)}, expected {(desc.batch, desc.c_out, l_out)}")
    dx = torch.empty((desc.batch, desc.c_in, desc.l_in), dtype=torch.float32, device=dy.device)
    for t, nm in ((add, "add"), (mask, "mask")):
        if t is not None and tuple(t.shape) != tuple(dx.shape):
            raise AgxError(f"conv_bwd_data: {nm} is {tuple(t.shape)}, dx is {tuple(dx.shape)}")
    add = None if add is None else _f32c(add)
    mask = None if mask is None else _f32c(mask)
    count_macs("conv_bwd_data", _conv_macs(desc), desc)
    _lib.check(lib.agx_conv_bwd_data(ctypes.byref(desc), _ptr(dy), _ptr(packed_bwd), _ptr(add), _ptr(mask),
                                     float(slope), _ptr(dx), _stream()), "agx_conv_bwd_data")
    return dx


def conv_bwd_weight(desc: ConvDesc, x: Tensor, dy: Tensor, v: Tensor, g: Optional[Tensor], want_bias: bool = True):
    """(dv, dg or None, dbias or None) of the layer ``desc`` describes."""
    lib = _lib.load()
    _need_gpu(x, dy, v, g)
    x, dy, v = _f32c(x), _f32c(dy), _f32c(v)
    g = None if g is None else _f32c(g)
    dv = torch.empty_like(v)
    dg = None if g is None else torch.empty_like(g)
    db = torch.empty(desc.c_out, dtype=torch.float32, device=x.device) if want_bias else None
    ws_bytes = int(lib.agx_conv_bwd_weight_workspace_bytes(ctypes.byref(desc)))
    ws = torch.empty(max(ws_bytes, 4) // 4, dtype=torch.float32, device=x.device)
    count_macs("conv_bwd_weight", _conv_macs(desc), desc)
    _lib.check(lib.agx_conv_bwd_weight(ctypes.byref(desc), _ptr(x), _ptr(dy), _ptr(v), _ptr(g), _ptr(dv), _ptr(dg),
                                       _ptr(db), _ptr(ws), ws_bytes, _stream()), "agx_conv_bwd_weight")
    return dv, dg, db


def resblock_forward(desc: ConvDesc, x: Tensor, packed1: Tensor, bias1: Optional[Tensor],
                     packed2: Tensor, bias2: Optional[Tensor], post_act: bool = True) -> Tensor:
    lib = _lib.load()
    _need_gpu(x, packed1, packed2, bias1, bias2)
    x = _f32c(x)
    if tuple(x.shape) != (desc.batch, desc.c_in, desc.l_in):
        raise AgxError(f"resblock_forward: x is {tuple(x.shape)}, descriptor says "
                       f"{(desc.batch, desc.c_in, desc.l_in)}")
    y = torch.empty_like(x)
    ws_bytes = int(lib.agx_resblock_workspace_bytes(ctypes.byref(desc)))
    ws = torch.empty(max(ws_bytes, 4) // 4, dtype=torch.float32, device=x.device)
    tok = _observer.begin("resblock", desc) if _observer is not None else None
    _lib.check(lib.agx_resblock_forward(ctypes.byref(desc), _ptr(x), _ptr(packed1), _ptr(bias1),
                                        _ptr(packed2), _ptr(bias2), _ptr(y), int(bool(post_act)),
                                        _ptr(ws), ws_bytes, _stream()), "agx_resblock_forward")
    if tok is not None:
        _observer.end(tok)
    return y


# ---------------------------------------------------------------------------- rvq
def rvq_pack(codebooks: Tensor, sizes: Optional[Sequence[int]] = None) -> Tensor:
    """Stage images of (Q, K, D) codebooks; ``sizes[q] <= K`` = real codewords of stage q (rows beyond are padding
    the search can never select) for quantizers with one codebook size per stage."""
    lib = _lib.load()
    _need_gpu(codebooks)
    codebooks = _f32c(codebooks)
    q, k, d = codebooks.shape
    n = lib.agx_rvq_packed_floats(q, k, d)
    if n < 0:
        _lib.check(int(n), "agx_rvq_packed_floats")
    packed = torch.empty(int(n), dtype=torch.float32, device=codebooks.device)
    if sizes is None:
        _lib.check(lib.agx_rvq_pack(_ptr(codebooks), q, k, d, _ptr(packed), _stream()), "agx_rvq_pack")
    else:
        if len(sizes) != q:
            raise AgxError(f"rvq_pack: {len(sizes)} sizes for {q} stages")
        arr = (ctypes.c_int32 * q)(*[int(v) for v in sizes])
        _lib.check(lib.agx_rvq_pack_sized(_ptr(codebooks), ctypes.cast(arr, ctypes.c_void_p), q, k, d, _ptr(packed),
                                          _stream()), "agx_rvq_pack_sized")
    return packed


def rvq_forward(x: Tensor, codebooks: Tensor, packed: Tensor, q_used: int,
                layout: str = "b l c") -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """x: (B,T,D) for layout "b l c" or (B,D,T) for "b c l" (any strides).
    Returns (x_q in the same layout/shape, index (B,T,q_used) int64, sq_err (q_used) f64, commit): the commit loss
    sum(sq_err) / x.numel() is a 0-d f32 tensor written by the same launch."""
    lib = _lib.load()
    _need_gpu(x, codebooks, packed)
    if x.dtype != torch.float32:
        raise AgxError(f"rvq_forward: expected float32, got {x.dtype}")
    codebooks = _f32c(codebooks)
    q_total, k, d = codebooks.shape
    if not 0 <= q_used <= q_total:
        raise AgxError(f"rvq_forward: q_used={q_used} outside [0, {q_total}]")
    if layout == "b l c":
        b, t, dim = x.shape
        sb, st, sd = x.stride()
    elif layout == "b c l":
        b, dim, t = x.shape
        sb, sd, st = x.stride()
    else:
        raise AgxError(f"rvq_forward: unknown layout {layout!r}")
    if dim != d:
        raise AgxError(f"rvq_forward: frame dim {dim} != codebook dim {d}")
    xq = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    if layout == "b l c":
        qb, qt, qd = xq.stride()
    else:
        qb, qd, qt = xq.stride()
    index = torch.empty((b, t, q_used), dtype=torch.int64, device=x.device)
    # one f64 buffer: [q_used per-stage sums | per-workgroup partials (the library's workspace)]; the commit loss is a
    # separate f32 scalar -- all three written by the launch, nothing to zero, no reduction on the torch side
    ws_bytes = int(lib.agx_rvq_workspace_bytes(b, t, d, k, q_used))
    buf = torch.empty(max(q_used, 1) + ws_bytes // 8, dtype=torch.float64, device=x.device)
    commit = torch.empty((), dtype=torch.float32, device=x.device)
    tok = _observer.begin("rvq", (b, t, d, k, q_used)) if _observer is not None else None
    _lib.check(lib.agx_rvq_forward_ex(_ptr(x), sb, st, sd, _ptr(codebooks), _ptr(packed), b, t, d, k, q_used,
                                      _ptr(xq), qb, qt, qd, _ptr(index), _ptr(buf), _ptr(commit),
                                      ctypes.c_void_p(buf.data_ptr() + 8 * max(q_used, 1)), ws_bytes, _stream()),
               "agx_rvq_forward_ex")
    if tok is not None:
        _observer.end(tok)
    if q_used == 0:
        xq.zero_()
        commit.zero_()
    return xq, index, buf[:q_used], commit


def rvq_ema_stats(frames: Tensor, codebooks: Tensor, index: Tensor) -> Tensor:
    """Per-stage, per-code assignment counts and residual sums of the EMA codebook update: frames (N, D),
    codebooks (Q, K, D), index (N, q_used) -> stats (q_used, K, D + 1); sums in frame order (deterministic)."""
    lib = _lib.load()
    _need_gpu(frames, codebooks, index)
    frames, codebooks = _f32c(frames), _f32c(codebooks)
    index = index.contiguous().to(torch.int64)
    n, d = frames.shape
    q_used = index.shape[1]
    k = codebooks.shape[1]
    stats = torch.empty(q_used, k, d + 1, dtype=torch.float32, device=frames.device)
    nbytes = lib.agx_rvq_ema_workspace_bytes(n, d, q_used)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=frames.device)
    _lib.check(lib.agx_rvq_ema_stats(_ptr(frames), _ptr(codebooks), _ptr(index), _ptr(stats), n, d, k, q_used, _ptr(ws), nbytes,
                                     _stream()), "agx_rvq_ema_stats")
    return stats


def rvq_dequantize(codebook: Tensor, idx: Tensor, out: Optional[Tensor] = None,
                   accumulate: bool = False) -> Tensor:
    """``codebook[idx]``: idx (...,) int64 -> (..., D)."""
    lib = _lib.load()
    _need_gpu(codebook, idx)
    codebook = _f32c(codebook)
    k, d = codebook.shape
    idx_c = idx.contiguous().to(torch.int64)
    n = idx_c.numel()
    if out is None:
        out = torch.empty((*idx.shape, d), dtype=torch.float32, device=codebook.device)
        accumulate = False
    flat = out.view(n, d)
    _lib.check(lib.agx_rvq_dequantize(_ptr(codebook), _ptr(idx_c), n, k, d, _ptr(flat), flat.stride(0),
                                      flat.stride(1), int(accumulate), _stream()), "agx_rvq_dequantize")
    return out


# ------------------------------------------------------------------ attention block
def layernorm_ct(x: Tensor, weight: Optional[Tensor], bias: Optional[Tensor], eps: float = 1e-5) -> Tensor:
    """LayerNorm over the channel dim of a (B, C, T) tensor."""
    lib = _lib.load()
    _need_gpu(x, weight, bias)
    x = _f32c(x)
    b, c, t = x.shape
    y = torch.empty_like(x)
    tok = _observer.begin("other", ("layernorm_ct", 8 * x.numel())) if _observer is not None else None
    _lib.check(lib.agx_layernorm_ct(_ptr(x), _ptr(None if weight is None else _f32c(weight)),
                                    _ptr(None if bias is None else _f32c(bias)), _ptr(y), b, c, t, float(eps),
                                    _stream()), "agx_layernorm_ct")
    if tok is not None:
        _observer.end(tok)
    return y


ATTN_FP32, ATTN_BF16 = 0, 1


def attention_alibi(qkv: Tensor, slopes: Tensor, heads: int, head_dim: int, scale_div: float,
                    precision: int = ATTN_FP32, flash: bool = False) -> Tensor:
    """softmax(QK^T/scale_div + ALiBi) V on a (B, 3*H*Dh, T) tensor -> (B, H*Dh, T), any T.
    ``precision``: ATTN_FP32 (exact fp32 MFMA) or ATTN_BF16 (bf16 MFMA, fp32 accumulate and softmax: BASELINE
    config 3); ``flash`` forces the online-softmax form for fp32 at T <= 256 as well."""
    lib = _lib.load()
    _need_gpu(qkv, slopes)
    qkv = _f32c(qkv)
    b, c3, t = qkv.shape
    if c3 != 3 * heads * head_dim:
        raise AgxError(f"attention_alibi: qkv has {c3} channels, expected {3 * heads * head_dim}")
    out = torch.empty((b, heads * head_dim, t), dtype=torch.float32, device=qkv.device)
    name = "attention_alibi" + (":bf16" if precision == ATTN_BF16 else "") + (":flash" if (flash or t > 256 or precision) else "")
    # algorithmic work: QK^T and PV, 2 * B * H * T * T * Dh MACs; bytes: qkv read once + out written once
    tok = (_observer.begin("other", (name, 4 * (qkv.numel() + out.numel()), 2 * b * heads * t * t * head_dim))
           if _observer is not None else None)
    _lib.check(lib.agx_attention_alibi_ex(_ptr(qkv), _ptr(_f32c(slopes)), _ptr(out), b, heads, head_dim, t,
                                          float(scale_div), int(precision), int(bool(flash)), _stream()),
               "agx_attention_alibi_ex")
    if tok is not None:
        _observer.end(tok)
    return out


def layernorm_ct_backward(x: Tensor, weight: Optional[Tensor], dy: Tensor, eps: float = 1e-5,
                          add: Optional[Tensor] = None):
    """(dx [+ add], dweight, dbias) of ``layernorm_ct``."""
    lib = _lib.load()
    _need_gpu(x, weight, dy, add)
    x, dy = _f32c(x), _f32c(dy)
    add = None if add is None else _f32c(add)
    b, c, t = x.shape
    dx = torch.empty_like(x)
    dw = torch.empty(c, dtype=torch.float32, device=x.device)
    db = torch.empty(c, dtype=torch.float32, device=x.device)
    ws = torch.empty(2 * b * ((t + 63) // 64) * c, dtype=torch.float32, device=x.device)
    _lib.check(lib.agx_layernorm_ct_backward(_ptr(x), _ptr(None if weight is None else _f32c(weight)), _ptr(dy),
                                             _ptr(add), _ptr(dx), _ptr(dw), _ptr(db), _ptr(ws), b, c, t, float(eps),
                                             _stream()), "agx_layernorm_ct_backward")
    return dx, dw, db


def attention_alibi_backward(qkv: Tensor, slopes: Tensor, dout: Tensor, heads: int, head_dim: int,
                             scale_div: float, out: Optional[Tensor] = None) -> Tensor:
    """dqkv of ``attention_alibi``.  T <= 256 and head_dim <= 64: the single-launch kernel; otherwise the flash-style
    split (``agx_attention_alibi_backward_ex``), which needs ``out`` = the forward's output."""
    lib = _lib.load()
    _need_gpu(qkv, slopes, dout, out)
    qkv, dout = _f32c(qkv), _f32c(dout)
    b, _, t = qkv.shape
    dqkv = torch.empty_like(qkv)
    if t <= 256 and head_dim <= 64:
        _lib.check(lib.agx_attention_alibi_backward(_ptr(qkv), _ptr(_f32c(slopes)), _ptr(dout), _ptr(dqkv), b, heads,
                                                    head_dim, t, float(scale_div), _stream()),
                   "agx_attention_alibi_backward")
        return dqkv
    if out is None:
        raise AgxError("attention_alibi_backward: T > 256 or head_dim > 64 needs the forward output (out=)")
    nbytes = lib.agx_attention_backward_workspace_bytes(b, heads, t)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=qkv.device)
    _lib.check(lib.agx_attention_alibi_backward_ex(_ptr(qkv), _ptr(_f32c(slopes)), _ptr(_f32c(out)), _ptr(dout), _ptr(dqkv),
                                                   _ptr(ws), nbytes, b, heads, head_dim, t, float(scale_div), _stream()),
               "agx_attention_alibi_backward_ex")
    return dqkv


def conv_bwd_data_gelu(desc: ConvDesc, dy: Tensor, packed_bwd: Tensor, pre: Tensor, add: Optional[Tensor] = None) -> Tensor:
    """``conv_bwd_data`` followed (in the epilogue) by the exact-GELU gradient at the pre-activation ``pre``."""
    lib = _lib.load()
    _need_gpu(dy, packed_bwd, pre, add)
    dy, pre = _f32c(dy), _f32c(pre)
    add = None if add is None else _f32c(add)
    dx = torch.empty(desc.batch, desc.c_in, desc.l_in, dtype=torch.float32, device=dy.device)
    count_macs("conv_bwd_data", _conv_macs(desc), desc)
    _lib.check(lib.agx_conv_bwd_data_gelu(ctypes.byref(desc), _ptr(dy), _ptr(packed_bwd), _ptr(add), _ptr(pre), _ptr(dx),
                                          _stream()), "agx_conv_bwd_data_gelu")
    return dx


# ------------------------------------------------------------------ wavelet layers
def multires_forward(x: Tensor, h0: Tensor, h1: Tensor, w: Tensor, depth: int) -> Tensor:
    lib = _lib.load()
    _need_gpu(x, h0, h1, w)
    x = _f32c(x)
    b, c, length = x.shape
    k = h0.shape[-1]
    if tuple(h0.shape) != (c, 1, k) or tuple(h1.shape) != (c, 1, k) or tuple(w.shape) != (c, depth + 2):
        raise AgxError("multires_forward: parameter shapes do not match (C,1,K) / (C,depth+2)")
    y = torch.empty_like(x)
    _lib.check(lib.agx_multires_forward(_ptr(x), _ptr(_f32c(h0)), _ptr(_f32c(h1)), _ptr(_f32c(w)), _ptr(y),
                                        b, c, length, k, depth, _stream()), "agx_multires_forward")
    return y


def multires_backward(x: Tensor, dout: Tensor, h0: Tensor, h1: Tensor, w: Tensor, depth: int):
    """(dx, dh0, dh1, dw) of ``multires_forward``."""
    lib = _lib.load()
    _need_gpu(x, dout, h0, h1, w)
    x, dout = _f32c(x), _f32c(dout)
    b, c, length = x.shape
    k = h0.shape[-1]
    if dout.shape != x.shape or tuple(h0.shape) != (c, 1, k) or tuple(h1.shape) != (c, 1, k) or tuple(w.shape) != (c, depth + 2):
        raise AgxError("multires_backward: shapes do not match (B,C,L) / (C,1,K) / (C,depth+2)")
    dx, dh0, dh1, dw = torch.empty_like(x), torch.empty_like(h0), torch.empty_like(h1), torch.empty_like(w)
    nbytes = int(lib.agx_multires_backward_workspace_bytes(b, c, length, k, depth))
    ws = torch.empty(max(nbytes, 4) // 4, dtype=torch.float32, device=x.device)
    _lib.check(lib.agx_multires_backward(_ptr(x), _ptr(dout), _ptr(_f32c(h0)), _ptr(_f32c(h1)), _ptr(_f32c(w)), _ptr(dx),
                                         _ptr(dh0), _ptr(dh1), _ptr(dw), _ptr(ws), nbytes, b, c, length, k, depth,
                                         _stream()), "agx_multires_backward")
    return dx, dh0, dh1, dw


def group_sum(g: Tensor, group: int, gelu_pre: Optional[Tensor] = None) -> Tensor:
    """out[..., l] = sum_{j < group} g[..., l * group + j]  (adjoint of a nearest-neighbour upsample), times the exact-GELU
    derivative at ``gelu_pre[..., l]`` when given."""
    lib = _lib.load()
    _need_gpu(g, gelu_pre)
    g = _f32c(g)
    length = g.shape[-1]
    if group <= 0 or length % group:
        raise AgxError(f"group_sum: last dim {length} is not a multiple of {group}")
    out = torch.empty(*g.shape[:-1], length // group, dtype=torch.float32, device=g.device)
    if gelu_pre is not None:
        gelu_pre = _f32c(gelu_pre)
        if gelu_pre.shape != out.shape:
            raise AgxError(f"group_sum: gelu_pre is {tuple(gelu_pre.shape)}, output is {tuple(out.shape)}")
    _lib.check(lib.agx_group_sum(_ptr(g), _ptr(gelu_pre), _ptr(out), out.numel(), group, _stream()), "agx_group_sum")
    return out


def wavelet_fold(h: Tensor, space: Tensor, sigma: Tensor, scale: int) -> Tensor:
    lib = _lib.load()
    _need_gpu(h, space, sigma)
    h = _f32c(h)
    b, c, length = h.shape
    space = _f32c(space.reshape(-1))
    sigma = _f32c(sigma.reshape(-1))
    y = torch.empty((b, c, length * scale), dtype=torch.float32, device=h.device)
    _lib.check(lib.agx_wavelet_fold(_ptr(h), _ptr(space), _ptr(sigma), sigma.numel(), _ptr(y), b, c, length,
                                    space.numel(), scale, _stream()), "agx_wavelet_fold")
    return y


def wavelet_fold_backward(h: Tensor, dout: Tensor, space: Tensor, sigma: Tensor, scale: int):
    """(dh, dsigma) of ``wavelet_fold``; dsigma has sigma's number of elements."""
    lib = _lib.load()
    _need_gpu(h, dout, space, sigma)
    h, dout = _f32c(h), _f32c(dout)
    b, c, length = h.shape
    space = _f32c(space.reshape(-1))
    sig = _f32c(sigma.reshape(-1))
    dh = torch.empty_like(h)
    dsig = torch.empty_like(sig)
    ws = torch.empty(b * c, dtype=torch.float32, device=h.device)
    _lib.check(lib.agx_wavelet_fold_backward(_ptr(h), _ptr(dout), _ptr(space), _ptr(sig), sig.numel(), _ptr(dh),
                                             _ptr(dsig), _ptr(ws), b, c, length, space.numel(), scale, _stream()),
               "agx_wavelet_fold_backward")
    return dh, dsig.reshape(sigma.shape)


# ------------------------------------------------------------------ discriminators
def spectral_sigma(w: Tensor, u: Tensor, v: Tensor, power_iterations: int, eps: float = 1e-12) -> Tensor:
    """sigma (1-element device tensor) of ``w`` viewed as (dim 0, rest); ``u`` / ``v`` are updated IN PLACE
    when ``power_iterations > 0`` (torch spectral_norm in training mode)."""
    lib = _lib.load()
    _need_gpu(w, u, v)
    w = _f32c(w)
    rows, cols = w.shape[0], w.numel() // w.shape[0]
    assert u.is_contiguous() and v.is_contiguous() and u.numel() == rows and v.numel() == cols
    sigma = torch.empty(1, dtype=torch.float32, device=w.device)
    ws = torch.empty(rows + cols, dtype=torch.float32, device=w.device)
    _lib.check(lib.agx_spectral_sigma(_ptr(w), rows, cols, _ptr(u), _ptr(v), power_iterations, eps, _ptr(sigma),
                                      _ptr(ws), _stream()), "agx_spectral_sigma")
    return sigma


def conv_pack_sigma(desc: ConvDesc, w: Tensor, sigma: Tensor) -> Tensor:
    lib = _lib.load()
    _need_gpu(w, sigma)
    n = lib.agx_conv_packed_floats(ctypes.byref(desc))
    if n < 0:
        _lib.check(int(n), "agx_conv_packed_floats")
    w = _f32c(w)
    packed = torch.empty(int(n), dtype=torch.float32, device=w.device)
    _lib.check(lib.agx_conv_pack_sigma(ctypes.byref(desc), _ptr(w), _ptr(sigma), _ptr(packed), _stream()),
               "agx_conv_pack_sigma")
    return packed


def conv_pack_bwd_sigma(desc: ConvDesc, w: Tensor, sigma: Tensor) -> Tensor:
    lib = _lib.load()
    _need_gpu(w, sigma)
    n = lib.agx_conv_bwd_packed_floats(ctypes.byref(desc))
    if n < 0:
        _lib.check(int(n), "agx_conv_bwd_packed_floats")
    w = _f32c(w)
    packed = torch.empty(int(n), dtype=torch.float32, device=w.device)
    _lib.check(lib.agx_conv_pack_bwd_sigma(ctypes.byref(desc), _ptr(w), _ptr(sigma), _ptr(packed), _stream()),
               "agx_conv_pack_bwd_sigma")
    return packed


def conv_grouped_bwd_data(desc: ConvDesc, dz: Tensor, w: Tensor, sigma: Optional[Tensor] = None,
                          add: Optional[Tensor] = None, mask: Optional[Tensor] = None, slope: float = 0.2) -> Tensor:
    lib = _lib.load()
    _need_gpu(dz, w, sigma, add, mask)
    dz, w = _f32c(dz), _f32c(w)
    add = None if add is None else _f32c(add)
    mask = None if mask is None else _f32c(mask)
    dx = torch.empty(desc.batch, desc.c_in, desc.l_in, dtype=torch.float32, device=dz.device)
    count_macs("conv_bwd_data", _conv_macs(desc), desc)
    _lib.check(lib.agx_conv_grouped_bwd_data(ctypes.byref(desc), _ptr(dz), _ptr(w), _ptr(sigma), _ptr(add), _ptr(mask),
                                             slope, _ptr(dx), _stream()), "agx_conv_grouped_bwd_data")
    return dx


def conv_grouped_bwd_weight(desc: ConvDesc, x: Tensor, dz: Tensor, want_bias: bool = True):
    """Plain (dw, dbias) of a grouped AGX_CONV_PADDED layer; dw has the torch layout (c_out, c_in / groups, K)."""
    lib = _lib.load()
    _need_gpu(x, dz)
    x, dz = _f32c(x), _f32c(dz)
    g = max(desc.groups, 1)
    dw = torch.empty(desc.c_out, desc.c_in // g, desc.kernel, dtype=torch.float32, device=x.device)
    db = torch.empty(desc.c_out, dtype=torch.float32, device=x.device) if want_bias else None
    nbytes = int(lib.agx_conv_grouped_bwd_weight_workspace_bytes(ctypes.byref(desc)))
    ws = torch.empty(nbytes // 4 + 1, dtype=torch.float32, device=x.device)
    count_macs("conv_bwd_weight", _conv_macs(desc), desc)
    _lib.check(lib.agx_conv_grouped_bwd_weight(ctypes.byref(desc), _ptr(x), _ptr(dz), _ptr(dw), _ptr(db), _ptr(ws),
                                               nbytes, _stream()), "agx_conv_grouped_bwd_weight")
    return dw, db


def avgpool1d(x: Tensor, kernel: int, stride: int, padding: int) -> Tensor:
    lib = _lib.load()
    _need_gpu(x)
    x = _f32c(x)
    l_in = x.shape[-1]
    l_out = lib.agx_avgpool1d_out_len(l_in, kernel, stride, padding)
    if l_out < 0:
        _lib.check(int(l_out), "agx_avgpool1d_out_len")
    rows = x.numel() // l_in
    y = torch.empty(*x.shape[:-1], int(l_out), dtype=torch.float32, device=x.device)
    _lib.check(lib.agx_avgpool1d(_ptr(x), _ptr(y), rows, l_in, kernel, stride, padding, _stream()), "agx_avgpool1d")
    return y


def conv2d_desc(batch, c_in, c_out, h_in, w_in, kh, kw, stride=(1, 1), padding=(0, 0), epilogue=0, slope=0.2,
                impl=IMPL_AUTO) -> _lib.Conv2dDesc:
    return _lib.Conv2dDesc(batch, c_in, c_out, h_in, w_in, kh, kw, stride[0], stride[1], padding[0], padding[1],
                           epilogue, slope, impl)


def conv2d_pack(desc, w: Tensor, sigma: Optional[Tensor] = None) -> Tensor:
    lib = _lib.load()
    _need_gpu(w, sigma)
    n = lib.agx_conv2d_packed_floats(ctypes.byref(desc))
    if n < 0:
        _lib.check(int(n), "agx_conv2d_packed_floats")
    w = _f32c(w)
    packed = torch.empty(int(n), dtype=torch.float32, device=w.device)
    _lib.check(lib.agx_conv2d_pack(ctypes.byref(desc), _ptr(w), _ptr(sigma), _ptr(packed), _stream()),
               "agx_conv2d_pack")
    return packed


def conv2d_forward(desc, x: Tensor, packed: Tensor, bias: Optional[Tensor]) -> Tensor:
    lib = _lib.load()
    _need_gpu(x, packed, bias)
    x = _f32c(x)
    ho, wo = ctypes.c_int32(), ctypes.c_int32()
    _lib.check(lib.agx_conv2d_out_shape(ctypes.byref(desc), ctypes.byref(ho), ctypes.byref(wo)), "agx_conv2d_out_shape")
    y = torch.empty(desc.batch, desc.c_out, ho.value, wo.value, dtype=torch.float32, device=x.device)
    bias = None if bias is None else _f32c(bias)
    tok = _observer.begin("other", ("conv2d:bf16x3" if desc.impl == _lib.IMPL_MFMA_BF16X3 else "conv2d", 4 * (x.numel() + y.numel()),
                                     _conv2d_macs(desc))) if _observer is not None else None
    _lib.check(lib.agx_conv2d_forward(ctypes.byref(desc), _ptr(x), _ptr(packed), _ptr(bias), _ptr(y), _stream()),
               "agx_conv2d_forward")
    if tok is not None:
        _observer.end(tok)
    return y


def conv2d_pack_bwd(desc, w: Tensor, sigma: Optional[Tensor] = None) -> Tensor:
    lib = _lib.load()
    _need_gpu(w, sigma)
    n = lib.agx_conv2d_bwd_packed_floats(ctypes.byref(desc))
    if n < 0:
        _lib.check(int(n), "agx_conv2d_bwd_packed_floats")
    w = _f32c(w)
    packed = torch.empty(int(n), dtype=torch.float32, device=w.device)
    _lib.check(lib.agx_conv2d_pack_bwd(ctypes.byref(desc), _ptr(w), _ptr(sigma), _ptr(packed), _stream()),
               "agx_conv2d_pack_bwd")
    return packed


def conv2d_bwd_data(desc, dy: Tensor, packed_bwd: Tensor, mask: Optional[Tensor] = None, slope: float = 0.2,
                    add: Optional[Tensor] = None) -> Tensor:
    """Gradient w.r.t. the input of the Conv2d layer ``desc`` describes (forward descriptor); ``add`` is summed
    in and the LeakyReLU gradient (``mask``) applied in the kernel's epilogue."""
    lib = _lib.load()
    _need_gpu(dy, packed_bwd, mask, add)
    dy = _f32c(dy)
    mask = None if mask is None else _f32c(mask)
    add = None if add is None else _f32c(add)
    dx = torch.empty(desc.batch, desc.c_in, desc.h_in, desc.w_in, dtype=torch.float32, device=dy.device)
    count_macs("conv2d_bwd_data", _conv2d_macs(desc), desc)
    _lib.check(lib.agx_conv2d_bwd_data(ctypes.byref(desc), _ptr(dy), _ptr(packed_bwd), _ptr(add), _ptr(mask), slope,
                                       _ptr(dx), _stream()), "agx_conv2d_bwd_data")
    return dx


def conv2d_bwd_weight(desc, x: Tensor, dy: Tensor, w: Optional[Tensor] = None, sigma: Optional[Tensor] = None,
                      u: Optional[Tensor] = None, v: Optional[Tensor] = None, want_bias: bool = True):
    """(dw, dbias or None); with ``sigma`` the spectral-norm chain rule is applied (dw w.r.t. weight_orig)."""
    lib = _lib.load()
    _need_gpu(x, dy, w, sigma, u, v)
    x, dy = _f32c(x), _f32c(dy)
    dw = torch.empty(desc.c_out, desc.c_in, desc.kh, desc.kw, dtype=torch.float32, device=x.device)
    db = torch.empty(desc.c_out, dtype=torch.float32, device=x.device) if want_bias else None
    nbytes = int(lib.agx_conv2d_bwd_weight_workspace_bytes(ctypes.byref(desc)))
    ws = torch.empty(nbytes // 4 + 1, dtype=torch.float32, device=x.device)
    w = None if w is None else _f32c(w)
    count_macs("conv2d_bwd_weight", _conv2d_macs(desc), desc)
    _lib.check(lib.agx_conv2d_bwd_weight(ctypes.byref(desc), _ptr(x), _ptr(dy), _ptr(w), _ptr(sigma), _ptr(u), _ptr(v),
                                         _ptr(dw), _ptr(db), _ptr(ws), nbytes, _stream()), "agx_conv2d_bwd_weight")
    return dw, db


def conv2d_bwd_data_fewchannels(desc, dy: Tensor, w: Tensor, sigma: Optional[Tensor] = None,
                                add: Optional[Tensor] = None) -> Tensor:
    """Backward-data of a stride-1 Conv2d with very few input channels (c_in * kw >= 8 rows on the MFMA tiles instead
    of c_in): auxiliary (kh x 1) conv over dy, then a column fold (include/agx.h: agx_conv2d_colsplit_weights)."""
    lib = _lib.load()
    _need_gpu(dy, w, sigma, add)
    dy, w = _f32c(dy), _f32c(w)
    add = None if add is None else _f32c(add)
    wp = torch.empty(desc.c_in * desc.kw, desc.c_out, desc.kh, 1, dtype=torch.float32, device=dy.device)
    _lib.check(lib.agx_conv2d_colsplit_weights(ctypes.byref(desc), _ptr(w), _ptr(sigma), _ptr(wp), _stream()),
               "agx_conv2d_colsplit_weights")
    aux = conv2d_desc(desc.batch, desc.c_out, desc.c_in * desc.kw, dy.shape[2], dy.shape[3], desc.kh, 1, (1, 1),
                      (desc.kh - 1 - desc.pad_h, 0))
    pbuf = conv2d_forward(aux, dy, conv2d_pack(aux, wp), None)
    dx = torch.empty(desc.batch, desc.c_in, desc.h_in, desc.w_in, dtype=torch.float32, device=dy.device)
    _lib.check(lib.agx_conv2d_colsum(ctypes.byref(desc), _ptr(pbuf), _ptr(add), _ptr(dx), _stream()), "agx_conv2d_colsum")
    return dx


def conv2d_kernel_name(desc) -> str:
    buf = ctypes.create_string_buffer(96)
    _lib.check(_lib.load().agx_conv2d_kernel_name(ctypes.byref(desc), buf, 96), "agx_conv2d_kernel_name")
    return buf.value.decode()


def conv2d_bwd_data_kernel_name(desc) -> str:
    buf = ctypes.create_string_buffer(96)
    _lib.check(_lib.load().agx_conv2d_bwd_data_kernel_name(ctypes.byref(desc), buf, 96), "agx_conv2d_bwd_data_kernel_name")
    return buf.value.decode()


_STFT_IMAGES = {}


def stft(x: Tensor, n_fft: int, normalized: bool = True) -> Tensor:
    """(B, L) -> (B, 2, T, n_fft): two-sided rectangular-window STFT, hop n_fft / 4, reflect-centred."""
    lib = _lib.load()
    _need_gpu(x)
    x = _f32c(x)
    b, length = x.shape
    t = lib.agx_stft_frames(length, n_fft)
    if t < 0:
        _lib.check(int(t), "agx_stft_frames")
    key = (n_fft, bool(normalized), x.device)
    if key not in _STFT_IMAGES:
        img = torch.empty(int(lib.agx_stft_packed_floats(n_fft)), dtype=torch.float32, device=x.device)
        _lib.check(lib.agx_stft_pack(n_fft, int(normalized), _ptr(img), _stream()), "agx_stft_pack")
        _STFT_IMAGES[key] = img
    ws = torch.empty(int(lib.agx_stft_workspace_bytes(b, length, n_fft)) // 4, dtype=torch.float32, device=x.device)
    y = torch.empty(b, 2, int(t), n_fft, dtype=torch.float32, device=x.device)
    tok = _observer.begin("other", ("stft", 4 * (x.numel() + y.numel()), b * 2 * n_fft * n_fft * int(t))) if _observer is not None else None
    _lib.check(lib.agx_stft_forward(_ptr(x), _ptr(_STFT_IMAGES[key]), _ptr(y), _ptr(ws), b, length, n_fft, _stream()),
               "agx_stft_forward")
    if tok is not None:
        _observer.end(tok)
    return y


def stft_backward(dy: Tensor, length: int, n_fft: int, normalized: bool = True) -> Tensor:
    """Adjoint of ``stft``: (B, 2, T, n_fft) -> (B, L)."""
    lib = _lib.load()
    _need_gpu(dy)
    dy = _f32c(dy)
    b = dy.shape[0]
    key = ("bwd", n_fft, bool(normalized), dy.device)
    if key not in _STFT_IMAGES:
        img = torch.empty(int(lib.agx_stft_packed_floats(n_fft)), dtype=torch.float32, device=dy.device)
        _lib.check(lib.agx_stft_pack_bwd(n_fft, int(normalized), _ptr(img), _stream()), "agx_stft_pack_bwd")
        _STFT_IMAGES[key] = img
    ws = torch.empty(int(lib.agx_stft_workspace_bytes(b, length, n_fft)) // 4, dtype=torch.float32, device=dy.device)
    dx = torch.empty(b, length, dtype=torch.float32, device=dy.device)
    count_macs("stft_backward", b * 2 * n_fft * n_fft * dy.shape[2])
    _lib.check(lib.agx_stft_backward(_ptr(dy), _ptr(_STFT_IMAGES[key]), _ptr(dx), _ptr(ws), b, length, n_fft, _stream()),
               "agx_stft_backward")
    return dx


def avgpool1d_backward(dy: Tensor, l_in: int, kernel: int, stride: int, padding: int,
                       add: Optional[Tensor] = None) -> Tensor:
    lib = _lib.load()
    _need_gpu(dy, add)
    dy = _f32c(dy)
    add = None if add is None else _f32c(add)
    rows = dy.numel() // dy.shape[-1]
    dx = torch.empty(*dy.shape[:-1], l_in, dtype=torch.float32, device=dy.device)
    _lib.check(lib.agx_avgpool1d_backward(_ptr(dy), _ptr(add), _ptr(dx), rows, l_in, kernel, stride, padding, _stream()),
               "agx_avgpool1d_backward")
    return dx


def sigmoid_backward(dy: Tensor, s: Tensor) -> Tensor:
    lib = _lib.load()
    _need_gpu(dy, s)
    dy, s = _f32c(dy), _f32c(s)
    dz = torch.empty_like(s)
    _lib.check(lib.agx_sigmoid_backward(_ptr(dy), _ptr(s), _ptr(dz), s.numel(), _stream()), "agx_sigmoid_backward")
    return dz


def spectral_grad_(g: Tensor, w: Tensor, sigma: Tensor, u: Tensor, v: Tensor) -> Tensor:
    """In place: plain weight gradient -> gradient w.r.t. weight_orig of a spectrally normalised layer."""
    lib = _lib.load()
    _need_gpu(g, w, sigma, u, v)
    assert g.is_contiguous() and g.dtype == torch.float32
    rows, cols = g.shape[0], g.numel() // g.shape[0]
    ws = torch.empty(rows, dtype=torch.float32, device=g.device)
    _lib.check(lib.agx_spectral_grad(_ptr(g), _ptr(_f32c(w)), _ptr(sigma), _ptr(u), _ptr(v), rows, cols, _ptr(ws),
                                     _stream()), "agx_spectral_grad")
    return g


REDUCE_MEAN, REDUCE_HINGE_REAL, REDUCE_HINGE_FAKE, REDUCE_L1, REDUCE_ABS_EPS = 0, 1, 2, 3, 4


def reduce_mean(x: Tensor, mode: int, y: Optional[Tensor] = None) -> Tensor:
    """One of the means of ``discriminator_generator_loss`` as a 0-d device tensor."""
    lib = _lib.load()
    _need_gpu(x, y)
    x = _f32c(x)
    y = None if y is None else _f32c(y)
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    ws = torch.empty(1024, dtype=torch.float32, device=x.device)
    _lib.check(lib.agx_reduce_mean(_ptr(x), _ptr(y), x.numel(), mode, _ptr(out), _ptr(ws), _stream()),
               "agx_reduce_mean")
    return out[0]


def reduce_mean_backward(x: Tensor, mode: int, grad: Tensor, y: Optional[Tensor] = None, want_dy: bool = False):
    lib = _lib.load()
    _need_gpu(x, y, grad)
    x = _f32c(x)
    y = None if y is None else _f32c(y)
    grad = _f32c(grad.reshape(1))
    dx = torch.empty_like(x)
    dy = torch.empty_like(x) if want_dy else None
    _lib.check(lib.agx_reduce_mean_backward(_ptr(x), _ptr(y), x.numel(), mode, _ptr(grad), _ptr(dx), _ptr(dy),
                                            _stream()), "agx_reduce_mean_backward")
    return dx, dy


def feature_means(x: Tensor, y: Tensor) -> Tensor:
    """(mean|x - y|, mean|x + 1e-3|) of one feature-matching term in one pass: a 2-element device tensor."""
    lib = _lib.load()
    _need_gpu(x, y)
    x, y = _f32c(x), _f32c(y)
    if x.shape != y.shape:
        raise AgxError(f"feature_means: shapes differ ({tuple(x.shape)} vs {tuple(y.shape)})")
    out = torch.empty(2, dtype=torch.float32, device=x.device)
    ws = torch.empty(2048, dtype=torch.float32, device=x.device)
    _lib.check(lib.agx_feature_means(_ptr(x), _ptr(y), x.numel(), _ptr(out), _ptr(ws), _stream()), "agx_feature_means")
    return out


def feature_means_backward(x: Tensor, y: Tensor, grad: Tensor, want_dx: bool = True, want_dy: bool = True):
    lib = _lib.load()
    _need_gpu(x, y, grad)
    x, y = _f32c(x), _f32c(y)
    grad = _f32c(grad.reshape(2))
    dx = torch.empty_like(x) if want_dx else None
    dy = torch.empty_like(x) if want_dy else None
    if dx is None and dy is None:
        return None, None
    _lib.check(lib.agx_feature_means_backward(_ptr(x), _ptr(y), x.numel(), _ptr(grad), _ptr(dx), _ptr(dy), _stream()),
               "agx_feature_means_backward")
    return dx, dy


def sigmoid(x: Tensor) -> Tensor:
    lib = _lib.load()
    _need_gpu(x)
    x = _f32c(x)
    y = torch.empty_like(x)
    _lib.check(lib.agx_sigmoid(_ptr(x), _ptr(y), x.numel(), _stream()), "agx_sigmoid")
    return y


# ------------------------------------------------------------------ bitstream
def codes_pack(index: Tensor, bits: int) -> Tensor:
    """(..,) int64 codes -> uint8 stream, `bits` bits per code (dense, little-endian)."""
    lib = _lib.load()
    _need_gpu(index)
    idx = index.contiguous().to(torch.int64)
    n = idx.numel()
    nbytes = lib.agx_codes_packed_bytes(n, bits)
    if nbytes < 0 or n == 0:
        raise AgxError(f"codes_pack: bad arguments (n={n}, bits={bits})")
    out = torch.empty(int(nbytes), dtype=torch.uint8, device=idx.device)
    _lib.check(lib.agx_codes_pack(_ptr(idx), n, bits, _ptr(out), _stream()), "agx_codes_pack")
    return out


def codes_unpack(stream: Tensor, n_codes: int, bits: int) -> Tensor:
    lib = _lib.load()
    _need_gpu(stream)
    if stream.dtype != torch.uint8 or stream.numel() < lib.agx_codes_packed_bytes(n_codes, bits):
        raise AgxError("codes_unpack: stream must be uint8 and hold ceil(n_codes*bits/8) bytes")
    out = torch.empty(n_codes, dtype=torch.int64, device=stream.device)
    _lib.check(lib.agx_codes_unpack(_ptr(stream.contiguous()), n_codes, bits, _ptr(out), _stream()), "agx_codes_unpack")
    return out
