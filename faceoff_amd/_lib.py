"""ctypes binding of libfaceoff_hip.so (C ABI: include/faceoff_hip.h).

The product path has NO fallback: if the shared library is missing or a call fails, this
module raises.  Build with `python -c "import __graft_entry__ as g; g.build()"` or
`make -C faceoff_amd/csrc`.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FACEOFF_HIP_LIB", os.path.join(_HERE, "libfaceoff_hip.so"))   # override: A/B kernel builds

ABI_VERSION = 102          # == FO_ABI_VERSION of include/faceoff_hip.h (tests/test_host_cpu.py reads the header); load() refuses any other library
FO_IN_RELU, FO_BIAS, FO_MASK, FO_ADD, FO_OUT_RELU, FO_DEPTH2SPACE, FO_OUT_F32 = 1, 2, 4, 8, 16, 32, 64


class FaceoffHipError(RuntimeError):
    pass


def kernel_source_sha16():
    """sha256 (first 16 hex digits) over the kernel sources (csrc/*.hip, *.inc, *.h, *.cpp, sorted by name): what profiles/ stamps
    a measurement with, so that a committed per-kernel number (profiles/pmc_traffic.json) can be tied to the code it was measured on."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(_HERE, "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".inc", ".h", ".cpp")):
            h.update(name.encode())
            with open(os.path.join(d, name), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


class ConvDesc(C.Structure):
    """Mirror of `fo_conv_desc` (include/faceoff_hip.h)."""
    _fields_ = [(n, C.c_int32) for n in (
        "N", "T", "Hin", "Win", "Hm", "Wm", "Hout", "Wout", "Cin", "Cout", "KD", "KH", "KW",
        "stride", "padD", "padH", "padW", "ostride", "ophH", "ophW",
        "ldIn", "ldOut", "ldMask", "ldAdd", "flags")]


_P = C.c_void_p
_I = C.c_int
_L = C.c_int64
_F = C.c_float
_D = C.POINTER(ConvDesc)


class ConvExtra(C.Structure):
    """Mirror of `fo_conv_extra` (include/faceoff_hip.h): the optional outputs / bit-plane mask of fo_conv_bf16_ex."""
    _fields_ = [("pooled", C.c_void_p), ("ldPooled", C.c_int32), ("pool_idx", C.c_void_p), ("mask_bits", C.c_void_p), ("out_bits", C.c_void_p),
                ("pooled_bits", C.c_void_p)]


_X = C.POINTER(ConvExtra)

# name -> (restype, argtypes); must list every symbol include/faceoff_hip.h declares
SIGNATURES = {
    "fo_version": (_I, []),
    "fo_comm_unique_id": (_I, [_P]),
    "fo_comm_init": (_I, [C.POINTER(C.c_void_p), _I, _I, _P, _I]),
    "fo_comm_set_stream": (_I, [_P, _P]),
    "fo_comm_rank": (_I, [_P]),
    "fo_comm_world": (_I, [_P]),
    "fo_comm_issued": (_L, [_P]),
    "fo_comm_allreduce_async": (_I, [_P, _P, _L, _P]),
    "fo_comm_broadcast_async": (_I, [_P, _P, _L, _I, _P]),
    "fo_comm_wait": (_I, [_P, _P]),
    "fo_comm_destroy": (_I, [_P]),
    "fo_last_error": (C.c_char_p, []),
    "fo_kernel_notes": (_I, [_I]),
    "fo_last_kernel": (C.c_char_p, []),
    "fo_device_info": (_I, [C.POINTER(C.c_int32)]),
    "fo_selftest_lane_moves": (_I, [_P, _P]),
    "fo_nchw_to_nhwc": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "fo_nchw2_to_nhwc8": (_I, [_P, _I, _P, _I, _P, _I, _I, _I, _P]),
    "fo_nhwc_to_nchw": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "fo_pack_conv": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "fo_pack_conv_dgrad": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "fo_pack_convT_k4s2": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "fo_pack_convT_k4s2_fused": (_I, [_P, _P, _I, _I, _I, _P]),
    "fo_pack_convT_k4s2_cells": (_I, [_P, _P, _I, _I, _I, _P]),
    "fo_pack_convT_k4s2_cells_n": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "fo_conv_igemm": (_I, [_D, _P, _P, _P, _P, _P, _P, _P]),
    "fo_conv_igemm_variant": (_I, [_D]),
    "fo_resblock_fwd": (_I, [_D, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "fo_resblock_bwd_conv3_ws_bytes": (_L, [_L]),
    "fo_resblock_bwd_conv3": (_I, [_L, _P, _I, _P, _I, _P, _P, _I, _P, _P, _P, _L, _P]),
    "fo_conv_igemm_banked": (_I, [_D, _P, _P, _P, _I, _P]),
    "fo_wino_gemm": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "fo_wino_gemm_split": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "fo_wino_gemm_split_ws_bytes": (_L, [_I, _I, _I, _I]),
    "fo_wino_wgrad_split_ws_bytes": (_L, [_I, _I, _I, _I, _I, _I]),
    "fo_wino_wgrad_split": (_I, [_P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _I, _P]),
    "fo_wgrad_banked_ws_bytes": (_L, [_D, _I]),
    "fo_conv_wgrad_banked": (_I, [_D, _P, _P, _P, _I, _I, _P, _L, _I, _P]),
    "fo_wino_gradout": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _P]),
    "fo_wino_gradout_bias_ws_bytes": (_L, [_I, _I, _I, _I, _I]),
    "fo_wino_gradout_bias": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _P, _P, _L, _P]),
    "fo_w42_filter": (_I, [_P, _P, _I, _I, _I, _P]),
    "fo_w42_input_cells": (_I, [_P, _I, _P, _I, _I, _I, _I, _L, _P]),
    "fo_w42_input_full": (_I, [_P, _I, _P, _I, _I, _I, _I, _L, _P]),
    "fo_w42_output": (_I, [_P, _L, _P, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P]),
    "fo_w42_output_cells": (_I, [_P, _L, _P, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P]),
    "fo_w42_gradout": (_I, [_P, _I, _P, _I, _I, _I, _I, _L, _P]),
    "fo_w42_gradout_bias_ws_bytes": (_L, [_I, _I, _I, _I]),
    "fo_w42_gradout_bias": (_I, [_P, _I, _P, _I, _I, _I, _I, _L, _P, _P, _L, _P]),
    "fo_w42_wgrad_out": (_I, [_P, _P, _I, _I, _P]),
    "fo_wino_wgrad_out": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "fo_wino_filter": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "fo_wino_input": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _P]),
    "fo_wino_output": (_I, [_P, _P, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "fo_wgrad_ws_bytes": (_L, [_D]),
    "fo_conv_wgrad": (_I, [_D, _P, _P, _P, _I, _I, _P, _P, _L, _P]),
    "fo_bias_grad": (_I, [_P, _P, _L, _I, _I, _I, _P, _P]),
    "fo_vq_prepare": (_I, [_P, _P, _P, _P]),
    "fo_vq_assign_ws_bytes": (_L, []),
    "fo_vq_assign": (_I, [_P, _I, _L, _P, _P, _P, _P, _I, _P, _P, _P]),
    "fo_vq_stats_ws_bytes": (_L, [_L]),
    "fo_vq_stats": (_I, [_P, _I, _L, _P, _P, _P, _P, _P]),
    "fo_vq_ema": (_I, [_P, _P, _P, _P, _P, _F, _F, _F, _P]),
    "fo_vq_bwd": (_I, [_P, _I, _P, _I, _P, _I, _P, _F, _P, _I, _L, _P]),
    "fo_vq_gather": (_I, [_P, _P, _P, _I, _L, _P]),
    "fo_mse_slice_fwd": (_I, [_P, _I, _P, _I, _I, _I, _I, _P, _P, _P]),
    "fo_mse_slice_bwd": (_I, [_P, _I, _P, _I, _I, _I, _I, _P, _F, _P, _I, _P]),
    "fo_mse_slice_fwd_bwd": (_I, [_P, _I, _P, _I, _I, _I, _I, _P, _F, _P, _I, _P, _P, _P]),
    "fo_lpips_prep": (_I, [_P, _I, _I, _P, _I, _I, _I, _P, _P, _P]),
    "fo_lpips_prep_bwd": (_I, [_P, _I, _P, _I, _L, _P, _P, _F, _P]),
    "fo_maxpool2_fwd": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "fo_maxpool2_bwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "fo_lpips_tap_ws_bytes": (_L, [_I, _I, _I]),
    "fo_lpips_tap_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "fo_lpips_tap_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "fo_pack_conv_bf16": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "fo_pack_conv_dgrad_bf16": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "fo_conv_igemm_bf16": (_I, [_D, _P, _P, _P, _P, _P, _P]),
    "fo_conv_igemm_bf16_pool": (_I, [_D, _P, _P, _P, _P, _P, _I, _P]),
    "fo_conv_igemm_bf16_pool_idx": (_I, [_D, _P, _P, _P, _P, _P, _I, _P, _P]),
    "fo_conv_bf16": (_I, [_D, _P, _P, _P, _P, _P, _P, _P]),
    "fo_conv_bf16_ex": (_I, [_D, _P, _P, _P, _P, _P, _P, _X, _P]),
    "fo_wgrad_bf16_ws_bytes": (_L, [_D]),
    "fo_conv_wgrad_bf16": (_I, [_D, _P, _P, _P, _I, _I, _P, _P, _L, _P]),
    "fo_bias_grad_bf16_ws_bytes": (_L, [_I]),
    "fo_bias_grad_bf16": (_I, [_P, _P, _L, _I, _I, _I, _P, _P]),
    "fo_f32_to_bf16": (_I, [_P, _L, _P, _L, _L, _I, _P]),
    "fo_bf16_to_f32": (_I, [_P, _L, _P, _L, _L, _I, _P]),
    "fo_nchw2_to_nhwc8_bf16": (_I, [_P, _I, _P, _I, _P, _I, _I, _I, _P]),
    "fo_vq_assign2": (_I, [_P, _I, _L, _P, _P, _P, _P, _I, _P, _P, _I, _P, _P, _P]),
    "fo_vq_bwd_bf16": (_I, [_P, _I, _P, _I, _P, _I, _P, _F, _P, _I, _L, _P]),
    "fo_lpips_prep_bf16": (_I, [_P, _I, _I, _P, _I, _I, _I, _P, _P, _P]),
    "fo_lpips_prep_bwd_bf16": (_I, [_P, _P, _I, _L, _P, _P, _F, _P]),
    "fo_maxpool2_fwd_bf16": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "fo_maxpool2_bwd_bf16": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "fo_maxpool2_fwd_idx_bf16": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "fo_maxpool2_bwd_idx_bf16": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "fo_lpips_tap_ws_bytes_bf16": (_L, [_I, _I, _I, _I]),
    "fo_lpips_tap_fwd_bf16": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "fo_lpips_tap_bwd_bf16": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "fo_vgg_conv1_fused_bf16": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "fo_lpips_tap_fwd_bwd_bf16": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "fo_lpips_tap_fwd_bwd_unpool_bf16": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "fo_adam_flat": (_I, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _F, _F, _P]),
    "fo_zero": (_I, [_P, _L, _P]),
    "fo_relu": (_I, [_P, _I, _P, _I, _L, _I, _P]),
    "fo_add": (_I, [_P, _I, _P, _I, _P, _I, _L, _I, _P]),
}

class ConvNdDesc(C.Structure):
    """Mirror of `fo_convnd_desc` (include/faceoff_hip.h)."""
    _fields_ = [(n, C.c_int32) for n in (
        "N", "Ds", "Hs", "Ws", "Cs", "ldS", "Dd", "Hd", "Wd", "Cd", "ldD", "KD", "KH", "KW", "sD", "sH", "sW", "pD", "pH", "pW",
        "ldMask", "flags")] + [("slope", C.c_float)]


_ND = C.POINTER(ConvNdDesc)
FO_OUT_LRELU, FO_MASK_LRELU, FO_KSPLIT = 64, 128, 256
SIGNATURES.update({
    "fo_pack_convnd": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "fo_convnd_ws_bytes": (C.c_int64, [_ND, _I]),
    "fo_convnd": (_I, [_ND, _I, _P, _P, _P, _P, _P, _P, C.c_int64, _P]),
    "fo_wgradnd_splits": (_I, [_ND]),
    "fo_disc_head_fwd": (_I, [_ND, _P, _P, _P, _P, _P]),
    "fo_disc_head_dgrad": (_I, [_ND, _P, _P, _P, _P]),
    "fo_disc_head_wgrad_ws_bytes": (_L, [_ND]),
    "fo_disc_head_wgrad": (_I, [_ND, _P, _P, _P, _I, _P, _L, _P]),
    "fo_wgradnd_ws_bytes": (C.c_int64, [_ND]),
    "fo_wgradnd": (_I, [_ND, _P, _P, _P, _I, _P, C.c_int64, _P]),
    "fo_instnorm_lrelu_fwd": (_I, [_P, _I, _P, _I, _L, _I, _F, _F, _P, _P, _F, _I, _P]),
    "fo_instnorm_lrelu_bwd": (_I, [_P, _I, _P, _I, _P, _P, _I, _L, _I, _F, _P]),
    "fo_instnorm_ws_bytes": (_L, [_I, _L, _I]),
    "fo_instnorm_lrelu_fwd_batch": (_I, [_P, _I, _P, _I, _I, _L, _I, _F, _F, _P, _P, _P, _F, _I, _P, _L, _P]),
    "fo_instnorm_lrelu_bwd_batch": (_I, [_P, _I, _P, _I, _P, _P, _I, _I, _L, _I, _F, _P, _L, _P]),
    "fo_avgpool3_fwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "fo_avgpool3_bwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "fo_disc_pairs": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P]),
    "fo_disc_pairs_bwd": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _P, _I, _F, _P]),
    "fo_affine_warp": (_I, [_P, _P, _I, _I, _I, _I, C.POINTER(C.c_float), _I, _P]),
    "fo_denorm_u8": (_I, [_P, _I, _I, _I, _P, _I, _I, _I, _I, _P]),
    "fo_warp_affine_u8": (_I, [_P, _P, _I, _I, _I, _I, C.POINTER(C.c_double), _I, _P]),
    "fo_resize_center_u8": (_I, [_P, _P, _I, _I, _I, _I, C.POINTER(C.c_double), _I, _P]),
    "fo_flip_u8": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "fo_u8_to_norm_nchw": (_I, [_P, _P, _I, _I, _I, _I, _I, C.c_float, C.c_float, _P]),
    "fo_space_to_depth2": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "fo_s2d_filter": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "fo_ralsgan": (_I, [_P, _I, _P, _I, _I, _F, _F, _F, _P, _P, _P, _P, _P]),
})

_lib = None


def load():
    """Load the library (once).  Raises FaceoffHipError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: its wheel bundles a HIP runtime, and the process must end up with ONE libamdhip64 -- the one torch's
    # allocator and streams live in.  Loading this library before torch binds it to the system runtime instead, and
    # launches then fail with "no ROCm-capable device is detected".
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise FaceoffHipError(
            f"{LIB_PATH} not found: the HIP extension is required (no CPU fallback). "
            "Build it with `make -C faceoff_amd/csrc` (hipcc --offload-arch=gfx950).")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the .so is stale
        fn.restype = res
        fn.argtypes = args
    got = lib.fo_version()
    if got != ABI_VERSION:
        raise FaceoffHipError(
            f"{LIB_PATH} reports ABI version {got}, this binding is written against {ABI_VERSION} (include/faceoff_hip.h FO_ABI_VERSION): "
            "argument lists differ between the two -- rebuild the library from this tree (`make -C faceoff_amd/csrc`).")
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().fo_last_error().decode(errors="replace")
        raise FaceoffHipError(f"{what} failed with code {rc}: {msg}")


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        check(rc, name)


_CU = None


def cu_count():
    """Compute units of the current device (fo_device_info), cached."""
    global _CU
    if _CU is None:
        out = (C.c_int32 * 3)()
        call("fo_device_info", out)
        _CU = int(out[0])
    return _CU
