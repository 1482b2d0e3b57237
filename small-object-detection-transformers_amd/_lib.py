"""ctypes binding of libsodt_hip.so (declarations mirror include/sodt_hip.h).

The product path has NO fallback: if the library is missing or cannot be loaded the
import raises, and every op raises on a non-zero status.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB_PATH = os.path.join(HERE, "libsodt_hip.so")
# SODT_LIB_PATH swaps in an experimental build (tools/exp/ab_build.sh) for A/B timing on one box.  It is a tools-only switch:
# overrides() reports it, ops.version() carries it, and bench.py refuses to run with it set.
LIB_PATH = os.environ.get("SODT_LIB_PATH") or DEFAULT_LIB_PATH
# environment switches that change WHICH native code runs or what it skips; the library itself reads none of them any more
# (round 4), the names stay listed so that bench.py also refuses the retired ones instead of silently ignoring a typo'd intent
DIAGNOSTIC_ENV = ("SODT_LIB_PATH", "SODT_HG_DBG", "SODT_WMSA_ONE_WAVE")


def overrides():
    """Diagnostic environment switches that are set in this process (empty for a product run)."""
    return {k: os.environ[k] for k in DIAGNOSTIC_ENV if os.environ.get(k)}

MAX_SEG = 9
F32, BF16 = 0, 1
EPI_BIAS, EPI_RESID, EPI_GELU_DUAL, EPI_DGELU = 1, 2, 4, 8
EPI_STATS, EPI_AFFINE_SILU, EPI_DETECT, EPI_OUT_F32 = 16, 32, 64, 128
EPI_GELU, EPI_DGELU_RC, EPI_RELU, EPI_DRELU = 256, 512, 2048, 4096      # (1024: retired, see include/sodt_hip.h)
STATS_REPL = 16      # SODT_STATS_REPL: replicas of the [2][N] f64 BatchNorm statistics buffer


class Seg(C.Structure):
    _fields_ = [("p", C.c_void_p), ("ld", C.c_int), ("klen", C.c_int), ("dy", C.c_int), ("dx", C.c_int),
                ("mul", C.c_int), ("shr", C.c_int), ("Hi", C.c_int), ("Wi", C.c_int)]


class ASpec(C.Structure):
    _fields_ = [("s", Seg * MAX_SEG), ("nseg", C.c_int), ("spatial", C.c_int), ("Ho", C.c_int), ("Wo", C.c_int)]


class GemmArgs(C.Structure):
    _fields_ = [("a", ASpec),
                ("W", C.c_void_p), ("ldw", C.c_int),
                ("C", C.c_void_p), ("ldc", C.c_int),
                ("C2", C.c_void_p), ("ldc2", C.c_int),
                ("bias", C.c_void_p),
                ("R", C.c_void_p), ("ldr", C.c_int), ("rmod", C.c_int),
                ("aux", C.c_void_p), ("ldaux", C.c_int),
                ("stats", C.c_void_p),
                ("scale", C.c_void_p), ("shift", C.c_void_p),
                ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("flags", C.c_int),
                ("oscatter", C.c_int), ("omul", C.c_int), ("ody", C.c_int), ("odx", C.c_int),
                ("OH", C.c_int), ("OW", C.c_int),
                ("det_na", C.c_int), ("det_no", C.c_int), ("det_hw", C.c_int)]


class GemmTnArgs(C.Structure):
    _fields_ = [("dY", C.c_void_p), ("ldy", C.c_int),
                ("x", ASpec),
                ("dW", C.c_void_p), ("lddw", C.c_int),
                ("dbias", C.c_void_p),
                ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
                ("kperm_c", C.c_int), ("kperm_t", C.c_int), ("splits", C.c_int),
                ("partial", C.c_void_p), ("partial_floats", C.c_long)]


class PrepDesc(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p),
                ("d0", C.c_int), ("d1", C.c_int), ("d2", C.c_int),
                ("p0", C.c_int), ("p1", C.c_int), ("p2", C.c_int),
                ("dst_ld", C.c_int), ("inner_ld", C.c_int)]


# name -> argtypes (all return int except sodt_version)
_P, _I, _L, _F = C.c_void_p, C.c_int, C.c_long, C.c_float
class Conv3Geo(C.Structure):
    """sodt_conv3_geo (include/sodt_hip.h): weight-row / input-pixel / output-pixel maps of the direct 3x3 kernels."""
    _fields_ = [("w_row_stride", C.c_int), ("w_row_off", C.c_int), ("in_mul", C.c_int), ("in_i", C.c_int), ("in_j", C.c_int),
                ("out_mul", C.c_int), ("out_i", C.c_int), ("out_j", C.c_int)]


SIGNATURES = {
    "sodt_nms_candidates": [_P, _I, _I, C.c_float, _I, _P, _P, _I, _P, _P],
    "sodt_nms_workspace_bytes": [_L, C.POINTER(C.c_size_t)],
    "sodt_nms_select": [_P, _I, _P, _L, C.c_float, _I, _P, C.c_size_t, _P, _P, _P, _P],
    "sodt_gemm_nt": [C.POINTER(GemmArgs), _I, _P],
    "sodt_gemm_tn": [C.POINTER(GemmTnArgs), _I, _P],
    "sodt_layernorm_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _P],
    "sodt_layernorm_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "sodt_window_attn_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "sodt_window_attn_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "sodt_wmsa_pack": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "sodt_wmsa_block_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "sodt_window_attn_bwd_wm": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "sodt_wmsa_block_bwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "sodt_frontend_fwd": [_P, _P, _L, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "sodt_frontend_bwd": [_P, _P, _L, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _L, _I, _P],
    "sodt_patch_embed4_fwd": [_P, _P, _L, _P, _P, _P, _I, _I, _P],
    "sodt_patch_embed4_bwd": [_P, _P, _L, _P, _P, _P, _I, _I, _P],
    "sodt_cross_attn_ln_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "sodt_cross_attn_ln_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "sodt_bn_finalize": [_P, _P, _P, _P, _L, _I, _F, _F, _P],
    "sodt_bn_affine": [_P, _P, _P, _P, _P, _I, _P],
    "sodt_bn_silu_fwd": [_P, _P, _P, _P, _P, _I, _L, _I, _I, _P],
    "sodt_bn_silu_bwd_reduce": [_P, _I, _P, _P, _P, _P, _P, _L, _I, _I, _P],
    "sodt_bn_silu_bwd_apply": [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _P],
    "sodt_copy_rows": [_P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "sodt_gather_sum_rows": [_P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "sodt_detect_unpermute": [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    "sodt_detect_decode": [_P, _P, _P, _I, _I, _I, _I, _I, _F, _P],
    "sodt_prep_weights": [_P, _I, _I, _I, _P],
    "sodt_transpose_f32": [_P, _P, _I, _I, _I, _P],
    "sodt_cast": [_P, _P, _L, _I, _I, _P],
    "sodt_batch_sum": [_P, _P, _I, _L, _I, _P],
    "sodt_memset_zero": [_P, _L, _P],
    "sodt_gemm_set_variant": [_I],
    "sodt_maxpool5_fwd": [_P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _P],
    "sodt_maxpool5_bwd": [_P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "sodt_yolo_loss_workspace_bytes": [_L, _I, _I, C.POINTER(C.c_size_t)],
    "sodt_yolo_loss": [_P, _P, _I, _P, _I, _I, _I, _I, _I, _F, _F, _F, _F, _F, _F, _F, _P, C.c_size_t, _P, _P, _P],
    "sodt_sgd_ema_step": [_P, _P, _P, _P, _P, _I, _P, _L, _I, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float),
                          _I, _F, _F, _P],
    "sodt_preprocess_u8": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "sodt_bilinear_up2_fwd": [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    "sodt_bilinear_up2_bwd": [_P, _I, _P, _P, _I, _I, _I, _I, _I, _P],
    "sodt_pixel_shuffle2": [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    "sodt_add_rows": [_P, _I, _I, _P, _I, _I, _L, _I, _I, _P],
    "sodt_nchw_f32_from_rows": [_P, _I, _P, _I, _I, _I, _I, _I, _P],
    "sodt_rows_from_nchw_f32": [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    "sodt_convmlp_compose": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P],
    "sodt_convmlp_border_fix": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "sodt_convmlp_border_sums": [_P, _P, _I, _I, _I, _I, _I, _P],
    "sodt_convmlp_decompose": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P],
    "sodt_mlp_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _P],
    "sodt_col_stats": [_P, _I, _P, _L, _I, _I, _P],
    "sodt_conv3x3_c64n8_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "sodt_conv3x3_c64n8_dgrad": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "sodt_conv3x3_c64n8_wgrad": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "sodt_conv3x3_c64_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _I, _P],
    "sodt_conv3x3_c64_wgrad": [_P, _P, _P, _P, _P, _I, _I, _I, _P, _I, _P],
    "sodt_debug_wmsa_stamps": [_P, _I],
    "sodt_debug_wmsa_hg_stamps": [_P, _I],
}

_lib = None


def load():
    """Load the library (once).  Raises if it is absent: there is no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python small-object-detection-transformers_amd/build.py` "
            "(hipcc --offload-arch=gfx950). The MI355X path has no fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = C.c_int
    lib.sodt_wmsa_pack_bytes.argtypes = [_I, _I, _I, _I]
    lib.sodt_wmsa_pack_bytes.restype = C.c_long
    lib.sodt_frontend_bwd_workspace_bytes.argtypes = [_I, _I]
    lib.sodt_frontend_bwd_workspace_bytes.restype = C.c_long
    lib.sodt_conv3x3_c64n8_wgrad_scratch_bytes.argtypes = []
    lib.sodt_conv3x3_c64n8_wgrad_scratch_bytes.restype = C.c_long
    lib.sodt_conv3x3_c64_wgrad_scratch_bytes.argtypes = []
    lib.sodt_conv3x3_c64_wgrad_scratch_bytes.restype = C.c_long
    lib.sodt_version.restype = C.c_char_p
    lib.sodt_version.argtypes = []
    _lib = lib
    return lib


def exported_symbols():
    return list(SIGNATURES.keys()) + ["sodt_version", "sodt_wmsa_pack_bytes", "sodt_frontend_bwd_workspace_bytes",
                                           "sodt_conv3x3_c64n8_wgrad_scratch_bytes", "sodt_conv3x3_c64_wgrad_scratch_bytes"]
