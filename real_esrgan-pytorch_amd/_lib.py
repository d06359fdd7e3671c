"""ctypes binding of csrc/libresr_hip.so (C-ABI declared in include/resr.h).

The product path has no CPU or PyTorch fallback: if the shared library is missing, `lib()` raises.
Structures mirror include/resr.h field for field.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RESR_LIB_PATH: experiment knob -- load a variant build (tools/build_variant.py) for same-box A/B timing
LIB_PATH = os.environ.get("RESR_LIB_PATH") or os.path.join(_HERE, "csrc", "libresr_hip.so")

RESR_F16, RESR_F32, RESR_F16X2 = 0, 1, 2
CONV_LRELU, CONV_UPSAMPLE_IN, CONV_CLAMP01, CONV_OUT_NCHW_F32, CONV_MASK, CONV_NO_BIAS = 1, 2, 4, 8, 16, 32
CONV_AUX_BEFORE_MASK, CONV_AUX_BEFORE_RES = 64, 128
CONV_WRITE_SIGNBITS = 1 << 8
CONV_MASK_BITS = 1 << 9
CONV_OUT_SINGLE = 1 << 10
CONV_SINGLE_W16 = 1 << 11
X2_PLAN_GROWTH_F16_INFER, X2_PLAN_GROWTH_GRAD_F16, X2_PLAN_GROWTH_GRAD_STORE_F16, X2_PLAN_GROWTH_ACT_F16_WGRAD = 1, 2, 4, 8
X2_PLAN_GROWTH_ACT_G_HI_WGRAD = 16
X2_PLAN_GROWTH_W16_INFER = 32
X2_PLAN_MX_INFER = 64
X2_PLAN_MX_BWD = 128
X2_PLAN_F16_BACKWARD = 256
X2_PLAN_MX_WGRAD = 512
X2_PLAN_MX_TAIL = 1024
CONV_MX_PAIRS = 1 << 12
RESR_VERSION = 3   # include/resr.h: the structures below mirror THIS version of the header


class ConvDesc(C.Structure):
    _fields_ = [("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32),
                ("cin", C.c_int32), ("cin0", C.c_int32), ("in0_stride", C.c_int32), ("in1_stride", C.c_int32),
                ("cout", C.c_int32), ("cout_pad", C.c_int32), ("out_stride", C.c_int32),
                ("res0_stride", C.c_int32), ("res1_stride", C.c_int32), ("mask_stride", C.c_int32),
                ("dtype", C.c_int32), ("flags", C.c_int32),
                ("s0", C.c_float), ("t0", C.c_float), ("s1", C.c_float), ("t1", C.c_float), ("slope", C.c_float),
                ("in0_chunk_stride", C.c_int32), ("in1_chunk_stride", C.c_int32), ("out_chunk_stride", C.c_int32),
                ("res0_chunk_stride", C.c_int32), ("res1_chunk_stride", C.c_int32), ("mask_chunk_stride", C.c_int32),
                ("in0_lo_offset", C.c_int64), ("in1_lo_offset", C.c_int64), ("out_lo_offset", C.c_int64),
                ("res0_lo_offset", C.c_int64), ("res1_lo_offset", C.c_int64),
                ("s2d_in_channels", C.c_int32), ("s2d_out_channels", C.c_int32), ("cout_groups", C.c_int32),
                ("x2_pair_chunks", C.c_int32), ("reserved2_", C.c_int32), ("mask_lo_offset", C.c_int64),
                ("in0_q_offset", C.c_int64), ("in1_q_offset", C.c_int64), ("out_q_offset", C.c_int64), ("w_mx_offset", C.c_int64)]


class WgradDesc(C.Structure):
    _fields_ = [("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32),
                ("cin", C.c_int32), ("cin0", C.c_int32), ("in0_stride", C.c_int32), ("in1_stride", C.c_int32),
                ("cin_real", C.c_int32), ("cout", C.c_int32), ("cout_pad", C.c_int32), ("g_stride", C.c_int32),
                ("dtype", C.c_int32), ("flags", C.c_int32), ("splits", C.c_int32), ("scale", C.c_float),
                ("x_lo_offset", C.c_int64), ("g_lo_offset", C.c_int64), ("x_chunk_stride", C.c_int64), ("g_chunk_stride", C.c_int64)]


class PackChunk(C.Structure):
    _fields_ = [("src_off", C.c_int64), ("dst_off", C.c_int64), ("src_cout", C.c_int32), ("src_cin", C.c_int32),
                ("m_off", C.c_int32), ("m_count", C.c_int32), ("k_off", C.c_int32), ("k_count", C.c_int32),
                ("mt", C.c_int32), ("transposed", C.c_int32), ("scale", C.c_float), ("virtual4x4", C.c_int32),
                ("scale_ptr", C.c_void_p)]


class ProfEntry(C.Structure):
    _fields_ = [("kernel_id", C.c_int32), ("ms", C.c_float), ("flop", C.c_double), ("bytes", C.c_double)]


class GeneratorDesc(C.Structure):
    _fields_ = [("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32),
                ("in_channels", C.c_int32), ("out_channels", C.c_int32), ("upscale", C.c_int32),
                ("n_blocks", C.c_int32), ("dtype", C.c_int32), ("training", C.c_int32), ("wgrad_splits", C.c_int32),
                ("x2_plan", C.c_int32), ("reserved_", C.c_int32)]


class DiscriminatorDesc(C.Structure):
    _fields_ = [("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("dtype", C.c_int32), ("training", C.c_int32),
                ("sn_training", C.c_int32)]


_P = C.c_void_p
_PROTOS = {
    "resr_discriminator_param_count": (C.c_size_t, []),
    "resr_discriminator_uv_count": (C.c_size_t, []),
    "resr_discriminator_workspace_bytes": (C.c_size_t, [C.POINTER(DiscriminatorDesc)]),
    "resr_discriminator_pack_table": (C.c_int64, [C.POINTER(DiscriminatorDesc), _P, _P, C.c_int64]),
    "resr_discriminator_forward": (C.c_int, [C.POINTER(DiscriminatorDesc), _P, _P, _P, _P, C.c_int32, _P, C.c_size_t, _P, _P]),
    "resr_discriminator_backward": (C.c_int, [C.POINTER(DiscriminatorDesc), _P, _P, _P, C.c_size_t, _P, _P, _P]),
    "resr_version": (C.c_int, []),
    "resr_last_error": (C.c_char_p, []),
    "resr_conv3x3": (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "resr_conv3x3_chain": (C.c_int, [C.c_int32, C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "resr_conv3x3_chain_state_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "resr_chain_errors": (C.c_int64, []),
    "resr_generator_chain_state_bytes": (C.c_size_t, [C.POINTER(GeneratorDesc)]),
    "resr_debug_occupy": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _P]),
    "resr_wgrad_partial_bytes": (C.c_size_t, [C.POINTER(WgradDesc)]),
    "resr_conv3x3_wgrad": (C.c_int, [C.POINTER(WgradDesc), _P, _P, _P, _P, _P, _P, _P]),
    "resr_pack_weights": (C.c_int, [_P, C.c_int32, _P, _P, C.c_int32, _P]),
    "resr_pack_weights_mx": (C.c_int, [_P, C.c_int32, _P, _P, _P]),
    "resr_generator_mx_offset": (C.c_size_t, [C.POINTER(GeneratorDesc)]),
    "resr_nchw_to_nhwc": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                    C.c_int32, _P, _P]),
    "resr_nhwc_to_nchw": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                    C.c_int32, _P]),
    "resr_sumpool2x2": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, _P]),
    "resr_generator_param_count": (C.c_size_t, [C.POINTER(GeneratorDesc)]),
    "resr_generator_packed_bytes": (C.c_size_t, [C.POINTER(GeneratorDesc), C.c_int32]),
    "resr_generator_workspace_bytes": (C.c_size_t, [C.POINTER(GeneratorDesc)]),
    "resr_generator_pack_table": (C.c_int64, [C.POINTER(GeneratorDesc), C.c_int32, _P, C.c_int64]),
    "resr_generator_buffer_offsets": (C.c_int64, [C.POINTER(GeneratorDesc), _P, C.c_int64]),
    "resr_generator_forward": (C.c_int, [C.POINTER(GeneratorDesc), _P, _P, _P, _P, C.c_size_t, _P, _P]),
    "resr_generator_backward": (C.c_int, [C.POINTER(GeneratorDesc), _P, _P, _P, _P, C.c_size_t, _P, _P, _P, _P, C.c_int32]),
    "resr_ema_update": (C.c_int, [_P, _P, C.c_int64, C.c_double, _P]),
    "resr_debug_tr_probe": (C.c_int, [_P, _P]),
    "resr_debug_conv_trace": (C.c_int, [_P]),
    "resr_debug_chain_errors": (C.c_int64, []),
    "resr_debug_wgrad_plan": (C.c_int, [C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_int32), C.c_int32]),
    "resr_debug_wgrad_dense_blocks": (C.c_int, [C.c_int32, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, C.c_size_t, _P, _P]),
    "resr_debug_sustained": (C.c_int, [C.c_int32, C.c_double, _P, C.c_size_t, _P, C.POINTER(C.c_double), C.POINTER(C.c_double), _P]),
    "resr_profile_begin": (C.c_int, []),
    "resr_profile_end": (C.c_int64, [_P, C.c_int64]),
    "resr_space_to_depth": (C.c_int, [_P, _P] + [C.c_int32] * 6 + [_P]),
    "resr_bilinear_up2x": (C.c_int, [_P, _P] + [C.c_int32] * 6 + [_P]),
    "resr_add_mask": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int32, C.c_float, _P]),
    "resr_l1_partial": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_int64, _P, C.c_int32, _P]),
    "resr_loss_scratch_bytes": (C.c_size_t, []),
    "resr_bce_logits_const": (C.c_int, [_P, C.c_int64, C.c_float, C.c_float, _P, _P, _P, _P]),
    "resr_l1_mean": (C.c_int, [_P, _P, C.c_int64, C.c_float, _P, _P, _P, _P]),
    "resr_weighted_row_sums": (C.c_int, [_P, C.c_int32, C.c_int32, C.POINTER(C.c_float), _P, _P]),
    "resr_spectral_norm": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_float, _P, _P, _P]),
    "resr_spectral_norm_bwd": (C.c_int, [_P] * 6 + [C.c_int32] * 3 + [_P, _P]),
    "resr_maxpool2x2": (C.c_int, [_P, _P] + [C.c_int32] * 5 + [_P]),
    "resr_maxpool2x2_arg": (C.c_int, [_P, _P, _P] + [C.c_int32] * 5 + [_P]),
    "resr_maxpool2x2_bwd": (C.c_int, [_P, _P, _P] + [C.c_int32] * 5 + [_P]),
    "resr_fold4x4": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P]),
    "resr_filter2d": (C.c_int, [_P, _P, _P] + [C.c_int32] * 7 + [_P]),
    "resr_usm_sharp": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_float, C.c_float] + [C.c_int32] * 4 + [_P]),
    "resr_usm_sharp_forward_only": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_float, C.c_float] + [C.c_int32] * 4 + [_P]),
    "resr_usm_sharp_bwd": (C.c_int, [_P] * 6 + [C.c_int32, C.c_float] + [C.c_int32] * 4 + [_P]),
    "resr_resize": (C.c_int, [_P, _P] + [C.c_int32] * 7 + [C.c_double, C.c_double, _P]),
    "resr_randn_fill": (C.c_int, [_P, C.c_int64, C.c_uint64, C.c_uint64, _P]),
    "resr_noise_gaussian": (C.c_int, [_P] * 6 + [C.c_int32] * 5 + [_P]),
    "resr_noise_poisson_workspace_bytes": (C.c_size_t, [C.c_int32]),
    "resr_noise_poisson": (C.c_int, [_P] * 4 + [C.c_uint64, _P] + [C.c_int32] * 5 + [_P]),
    "resr_jpeg": (C.c_int, [_P] * 4 + [C.c_int32] * 4 + [_P]),
    "resr_quantize_crop": (C.c_int, [_P] * 4 + [C.c_int32] * 10 + [_P]),
    "resr_filter2d_u8": (C.c_int, [_P, _P, _P] + [C.c_int32] * 7 + [_P]),
    "resr_resize_u8": (C.c_int, [_P, _P] + [C.c_int32] * 7 + [_P] * 5),
    "resr_jpeg_u8": (C.c_int, [_P, _P, _P, _P] + [C.c_int32] * 3 + [_P]),
}

_lib = None


def lib() -> C.CDLL:
    """Load libresr_hip.so (built by csrc/build.py).  Raises if it is missing -- there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python real_esrgan-pytorch_amd/csrc/build.py` "
                "(or __graft_entry__.build()); the MI355X path has no CPU/PyTorch fallback")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        if handle.resr_version() != RESR_VERSION:
            raise RuntimeError(f"{LIB_PATH} reports ABI version {handle.resr_version()}, this package binds version {RESR_VERSION}: "
                               "rebuild it (python real_esrgan-pytorch_amd/csrc/build.py --force)")
        _lib = handle
    return _lib


def exported_symbols():
    return sorted(_PROTOS)


def check(rc: int, what: str = "resr") -> None:
    if rc != 0:
        msg = lib().resr_last_error()
        raise RuntimeError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


def chain_health(sync: bool = False) -> None:
    """Raise if a chained dense-block launch (conv3x3_ws.h, CH) ever gave up waiting for a neighbouring tile or found
    an XCD with more than its share of workgroups (such a launch also poisons its output with NaNs, so a training loss
    shows it at once).  sync=False reads two host-mapped counters -- no synchronisation, cheap enough for every logging
    interval; sync=True drains the device first (end of an epoch / a benchmark)."""
    e = int(lib().resr_debug_chain_errors() if sync else lib().resr_chain_errors())
    if e != 0:
        raise RuntimeError(f"chained conv launches reported errors: {e & 0xffffffff} polls timed out, {e >> 32} workgroups beyond "
                           "their XCD's share (set RESR_CONV_NO_CHAIN=1 to run one launch per pass)")


def ptr(t):
    """Raw device pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr(ref=None):
    """The HIP stream a call should be enqueued on: the current stream of the device that owns `ref` (a tensor), else of
    the current device.  The library switches to the stream's device itself (csrc/api.hip DeviceScope)."""
    import torch
    dev = ref.device if ref is not None and getattr(ref, "is_cuda", False) else None
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def require_cuda(t, what: str):
    if not t.is_cuda:
        raise RuntimeError(f"{what}: expected a tensor on the MI355X device, got {t.device}; "
                           "this package has no CPU path (the CPU oracle lives under oracle/ for tests only)")
