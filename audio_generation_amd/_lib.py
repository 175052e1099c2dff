"""ctypes binding of ``libagx.so`` (the C ABI declared in ``include/agx.h``).

The product path has no other implementation: if the shared library is missing
or fails to load, importing an op raises -- there is no eager/PyTorch or CPU
fallback.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import (POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_int64,
                    c_size_t, c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libagx.so")

# constants mirrored from include/agx.h
CONV_CAUSAL, CONV_TRANSPOSED, CONV_UPSAMPLE, CONV_SAME, CONV_PADDED = 0, 1, 2, 3, 4
IMPL_AUTO, IMPL_DIRECT, IMPL_MFMA, IMPL_MFMA_BF16X3 = 0, 1, 2, 3
EPI_LEAKY_PRE, EPI_RESIDUAL, EPI_LEAKY_POST, EPI_GELU_PRE, EPI_MASK = 1, 2, 4, 8, 16


class ConvDesc(Structure):
    """``agx_conv_desc`` (include/agx.h)."""
    _fields_ = [("kind", c_int32), ("batch", c_int32), ("c_in", c_int32), ("c_out", c_int32),
                ("l_in", c_int32), ("kernel", c_int32), ("stride", c_int32), ("dilation", c_int32),
                ("epilogue", c_int32), ("slope", c_float), ("impl", c_int32), ("groups", c_int32),
                ("padding", c_int32)]


class Conv2dDesc(Structure):
    """``agx_conv2d_desc`` (include/agx.h)."""
    _fields_ = [("batch", c_int32), ("c_in", c_int32), ("c_out", c_int32), ("h_in", c_int32), ("w_in", c_int32),
                ("kh", c_int32), ("kw", c_int32), ("stride_h", c_int32), ("stride_w", c_int32),
                ("pad_h", c_int32), ("pad_w", c_int32), ("epilogue", c_int32), ("slope", c_float),
                ("impl", c_int32)]


# every symbol include/agx.h declares: name -> (restype, argtypes)
_PD = POINTER(ConvDesc)
_P2 = POINTER(Conv2dDesc)
SIGNATURES = {
    "agx_version": (c_int, []),
    "agx_last_error": (c_char_p, []),
    "agx_sizeof_conv_desc": (c_int32, []),
    "agx_sizeof_conv2d_desc": (c_int32, []),
    "agx_set_tuning": (c_int, [c_char_p, c_int32]),
    "agx_get_tuning": (c_int, [c_char_p]),
    "agx_conv_out_len": (c_int64, [_PD]),
    "agx_conv_packed_floats": (c_int64, [_PD]),
    "agx_conv_pack": (c_int, [_PD, c_void_p, c_void_p, c_void_p, c_void_p]),
    "agx_conv_forward": (c_int, [_PD, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "agx_planes_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "agx_planes_split": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "agx_conv_planes_supported": (c_int, [_PD]),
    "agx_conv_forward_planes": (c_int, [_PD, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "agx_conv_bwd_packed_floats": (c_int64, [_PD]),
    "agx_conv_pack_bwd": (c_int, [_PD, c_void_p, c_void_p, c_void_p, c_void_p]),
    "agx_conv_bwd_data": (c_int, [_PD, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p]),
    "agx_conv_bwd_weight_workspace_bytes": (c_size_t, [_PD]),
    "agx_conv_bwd_weight": (c_int, [_PD, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_size_t, c_void_p]),
    "agx_conv_kernel_name": (c_int, [_PD, c_char_p, c_size_t]),
    "agx_resblock_workspace_bytes": (c_size_t, [_PD]),
    "agx_resblock_kernel_name": (c_int, [_PD, c_char_p, c_size_t]),
    "agx_resblock_forward": (c_int, [_PD, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_int32, c_void_p, c_size_t, c_void_p]),
    "agx_rvq_packed_floats": (c_int64, [c_int32, c_int32, c_int32]),
    "agx_rvq_pack": (c_int, [c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "agx_rvq_pack_sized": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "agx_rvq_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32, c_int32, c_int32]),
    "agx_rvq_forward": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p,
                                c_int32, c_int32, c_int32, c_int32, c_int32,
                                c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p,
                                c_void_p, c_size_t, c_void_p]),
    "agx_rvq_forward_ex": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p,
                                   c_int32, c_int32, c_int32, c_int32, c_int32,
                                   c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_size_t, c_void_p]),
    "agx_rvq_ema_workspace_bytes": (c_size_t, [c_int64, c_int32, c_int32]),
    "agx_rvq_ema_stats": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_void_p,
                                  c_size_t, c_void_p]),
    "agx_rvq_debug_stamps": (c_int, [c_void_p, c_int32]),
    "agx_rvq_verify_counts": (c_int, [POINTER(c_int64), c_int32]),
    "agx_rvq_dequantize": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p,
                                   c_int64, c_int64, c_int32, c_void_p]),
    "agx_layernorm_ct": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_float,
                                 c_void_p]),
    "agx_attention_alibi": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_float,
                                    c_void_p]),
    "agx_attention_alibi_ex": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_float,
                                       c_int32, c_int32, c_void_p]),
    "agx_layernorm_ct_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_int32, c_int32, c_int32, c_float, c_void_p]),
    "agx_attention_alibi_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32,
                                             c_int32, c_float, c_void_p]),
    "agx_attention_backward_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "agx_attention_alibi_backward_ex": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                                c_int32, c_int32, c_int32, c_int32, c_float, c_void_p]),
    "agx_conv_bwd_data_gelu": (c_int, [_PD, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "agx_multires_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32,
                                     c_int32, c_int32, c_void_p]),
    "agx_multires_backward_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32, c_int32, c_int32]),
    "agx_multires_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_size_t, c_int32, c_int32, c_int32, c_int32, c_int32,
                                      c_void_p]),
    "agx_group_sum": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p]),
    "agx_wavelet_fold": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32,
                                 c_int32, c_int32, c_void_p]),
    "agx_wavelet_fold_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p,
                                          c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "agx_spectral_sigma": (c_int, [c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_int32, c_float, c_void_p,
                                   c_void_p, c_void_p]),
    "agx_conv_pack_sigma": (c_int, [_PD, c_void_p, c_void_p, c_void_p, c_void_p]),
    "agx_conv_grouped_bwd_data": (c_int, [_PD, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p,
                                          c_void_p]),
    "agx_conv_grouped_bwd_weight_workspace_bytes": (c_size_t, [_PD]),
    "agx_conv_grouped_bwd_weight": (c_int, [_PD, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "agx_conv_pack_bwd_sigma": (c_int, [_PD, c_void_p, c_void_p, c_void_p, c_void_p]),
    "agx_avgpool1d_out_len": (c_int64, [c_int32, c_int32, c_int32, c_int32]),
    "agx_avgpool1d": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "agx_conv2d_out_shape": (c_int, [_P2, POINTER(c_int32), POINTER(c_int32)]),
    "agx_conv2d_packed_floats": (c_int64, [_P2]),
    "agx_conv2d_pack": (c_int, [_P2, c_void_p, c_void_p, c_void_p, c_void_p]),
    "agx_conv2d_forward": (c_int, [_P2, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "agx_conv2d_kernel_name": (c_int, [_P2, c_char_p, c_size_t]),
    "agx_conv2d_bwd_data_kernel_name": (c_int, [_P2, c_char_p, c_size_t]),
    "agx_conv2d_bwd_packed_floats": (c_int64, [_P2]),
    "agx_conv2d_pack_bwd": (c_int, [_P2, c_void_p, c_void_p, c_void_p, c_void_p]),
    "agx_conv2d_bwd_data": (c_int, [_P2, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p]),
    "agx_conv2d_bwd_weight_workspace_bytes": (c_size_t, [_P2]),
    "agx_conv2d_bwd_weight": (c_int, [_P2, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_size_t, c_void_p]),
    "agx_conv2d_colsplit_weights": (c_int, [_P2, c_void_p, c_void_p, c_void_p, c_void_p]),
    "agx_conv2d_colsum": (c_int, [_P2, c_void_p, c_void_p, c_void_p, c_void_p]),
    "agx_stft_frames": (c_int64, [c_int32, c_int32]),
    "agx_stft_packed_floats": (c_int64, [c_int32]),
    "agx_stft_pack": (c_int, [c_int32, c_int32, c_void_p, c_void_p]),
    "agx_stft_workspace_bytes": (c_int64, [c_int32, c_int32, c_int32]),
    "agx_stft_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "agx_stft_pack_bwd": (c_int, [c_int32, c_int32, c_void_p, c_void_p]),
    "agx_stft_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "agx_avgpool1d_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32,
                                       c_void_p]),
    "agx_sigmoid_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "agx_spectral_grad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p,
                                  c_void_p]),
    "agx_reduce_mean": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p]),
    "agx_reduce_mean_backward": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "agx_feature_means": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "agx_feature_means_backward": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "agx_sigmoid": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "agx_fdft_frames": (c_int64, [c_int32, c_int32, c_int32]),
    "agx_fdft_rows": (c_int64, [c_int32, c_int32]),
    "agx_fdft_packed_floats": (c_int64, [c_int32, c_int32, c_int32, c_int32, c_int32]),
    "agx_fdft_pack": (c_int, [c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "agx_fdft_workspace_bytes": (c_int64, [c_int32, c_int32, c_int32, c_int32]),
    "agx_fdft_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32,
                                 c_int32, c_void_p]),
    "agx_fdft_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32,
                                  c_int32, c_void_p]),
    "agx_melpower": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "agx_melpower_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                      c_void_p]),
    "agx_preemphasis": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_float, c_int32, c_void_p]),
    "agx_lowpass_biquad": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_float, c_float, c_float, c_void_p]),
    "agx_resample_out_len": (c_int64, [c_int64, c_int32, c_int32]),
    "agx_resample": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "agx_codes_packed_bytes": (c_int64, [c_int64, c_int32]),
    "agx_codes_pack": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_void_p]),
    "agx_codes_unpack": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_void_p]),
}

_lib = None


class AgxError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load libagx.so once; raise loudly when it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AgxError(
            f"{LIB_PATH} not found: build it with `python -m audio_generation_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no fallback path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI and the header drifted apart
        fn.restype = res
        fn.argtypes = args
    if (lib.agx_sizeof_conv_desc() != ctypes.sizeof(ConvDesc)
            or lib.agx_sizeof_conv2d_desc() != ctypes.sizeof(Conv2dDesc)):
        raise AgxError("libagx.so was built from a different include/agx.h than this binding (descriptor sizes differ): "
                       "rebuild with `python -m audio_generation_amd.build`")
    for kv in filter(None, os.environ.get("AGX_TUNING", "").split(",")):     # measurement knobs, e.g. AGX_TUNING=dw_wgs=768
        name, _, value = kv.partition("=")
        if lib.agx_set_tuning(name.strip().encode(), int(value)) != 0:
            raise AgxError(f"AGX_TUNING: unknown knob {name!r}")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().agx_last_error()
        raise AgxError(f"{what} failed ({rc}): {msg.decode() if msg else '?'}")


def needs_grad(x, module) -> bool:
    """True when a forward must record what its backward kernels need (autograd on, and the input or a parameter of
    ``module`` requires a gradient)."""
    import torch
    return torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in module.parameters()))
