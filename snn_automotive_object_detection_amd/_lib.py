"""ctypes binding of libsnnhip.so (C ABI: include/snn_hip.h).  There is NO fallback: if the HIP
library is missing or a call fails, the product path raises."""
import ctypes as C
import os

from . import build as _build

c_f32p = C.POINTER(C.c_float)
c_u32p = C.POINTER(C.c_uint32)
c_u64p = C.POINTER(C.c_ulonglong)
c_stream = C.c_void_p

SNN_MAX_LEVELS = 8
SNN_MAX_STEPS = 32
PRECISIONS = {"f32": 0, "bf16x3": 1, "mxfp6": 2, "f32_strict": 3}


class snn_params(C.Structure):
    _fields_ = [("dt_tau_mem", C.c_float), ("neg_dt_tau_syn", C.c_float), ("v_leak", C.c_float),
                ("v_reset", C.c_float), ("v_th_enc", C.c_float), ("v_th_lif", C.c_float),
                ("li_order", C.c_int32), ("precision", C.c_int32)]


class snn_rpn_level(C.Structure):
    _fields_ = [("feat", C.c_void_p), ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("precision", C.c_int32)]


class snn_roi_level(C.Structure):
    _fields_ = [("feat", C.c_void_p), ("H", C.c_int32), ("W", C.c_int32), ("spatial_scale", C.c_float),
                ("reserved", C.c_int32)]


class snn_rpn_post_level(C.Structure):
    _fields_ = [("logits", C.c_void_p), ("deltas", C.c_void_p), ("H", C.c_int32), ("W", C.c_int32),
                ("stride_h", C.c_float), ("stride_w", C.c_float), ("base_anchors", (C.c_float * 4) * 16)]


# every symbol include/snn_hip_debug.h declares (test plumbing of the same .so, not the drop-in boundary)
DEBUG_SYMBOLS = {
    "snn_debug_reload_knobs": (None, []),
    "snn_debug_encoder_thresholds": (C.c_int, [C.POINTER(snn_params), C.POINTER(C.c_float)]),
    "snn_debug_last_conv_path": (C.c_int, []),
    "snn_debug_last_fc6_path": (C.c_int, []),
    "snn_debug_last_det_planes": (None, [C.POINTER(C.c_uint64)]),
    "snn_debug_last_rpn_planes": (None, [C.POINTER(C.c_uint64)]),
    "snn_debug_tile_shape": (C.c_int, [C.c_int, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32)]),
    "snn_debug_clock_probe": (C.c_int, [C.c_void_p, C.c_uint, c_stream]),
}

# every symbol include/snn_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "snn_version": (C.c_int, []),
    "snn_last_error": (C.c_char_p, []),
    "snn_packed_gemm_elems": (C.c_size_t, [C.c_int, C.c_int]),
    "snn_packed_conv3x3_elems": (C.c_size_t, [C.c_int, C.c_int]),
    "snn_pack_conv3x3_weight": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, c_stream]),
    "snn_packed_linear_elems": (C.c_size_t, [C.c_int, C.c_int]),
    "snn_pack_linear_weight": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, c_stream]),
    "snn_packed_heads_elems": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "snn_pack_heads_weight": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, c_stream]),
    "snn_rpn_head_workspace_bytes": (C.c_size_t, [C.POINTER(snn_rpn_level), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "snn_rpn_head_forward": (C.c_int, [C.POINTER(snn_rpn_level), C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.POINTER(snn_params), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, c_stream]),
    "snn_rpn_head_forward_stages": (C.c_int, [C.POINTER(snn_rpn_level), C.c_int, C.c_int, C.c_int, C.c_int,
                                              C.POINTER(snn_params), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int,
                                              c_stream]),
    "snn_det_head_workspace_bytes": (C.c_size_t, [C.c_int] * 7),
    "snn_det_head_forward": (C.c_int, [C.c_void_p] + [C.c_int] * 6 + [C.POINTER(snn_params)] +
                             [C.c_void_p] * 10 + [C.c_size_t, c_stream]),
    "snn_rpn_rates_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "snn_rpn_rates": (C.c_int, [C.POINTER(snn_rpn_level), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_void_p, C.c_size_t, c_stream]),
    "snn_det_rates": (C.c_int, [C.c_int] * 7 + [C.c_void_p] * 5 + [c_stream]),
    "snn_affine_act_nchw": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p, c_stream]),
    "snn_encode_nchw": (C.c_int, [C.c_void_p] + [C.c_int] * 5 + [C.POINTER(snn_params), C.c_void_p, C.c_size_t, c_stream]),
    "snn_encode_rows": (C.c_int, [C.c_void_p] + [C.c_int] * 3 + [C.POINTER(snn_params), C.c_void_p, C.c_size_t, c_stream]),
    "snn_conv3x3_lif": (C.c_int, [C.c_void_p, C.c_size_t] + [C.c_int] * 6 + [C.POINTER(snn_params), C.c_void_p,
                                  C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, c_stream]),
    "snn_spike_gemm": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, c_stream]),
    "snn_lif_scan": (C.c_int, [C.c_void_p] + [C.c_int] * 4 + [C.POINTER(snn_params), C.c_void_p, C.c_size_t,
                               C.c_void_p, c_stream]),
    "snn_li_heads": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                               C.POINTER(snn_params), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, c_stream]),
    "snn_nms_workspace_bytes": (C.c_size_t, [C.c_int]),
    "snn_rpn_proposals_candidates": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "snn_rpn_proposals_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "snn_rpn_proposals": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float,
                                    C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                    c_stream]),
    "snn_det_postprocess_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "snn_det_postprocess": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float,
                                      C.c_float, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, c_stream]),
    "snn_det_exchange_payload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                           c_stream]),
    "snn_nms_sorted": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_size_t, c_stream]),
    "snn_roi_align_encode": (C.c_int, [C.POINTER(snn_roi_level), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_int, C.c_int, C.POINTER(snn_params), C.c_void_p, C.c_size_t, C.c_void_p, c_stream]),
    "snn_det_head_forward_roialign": (C.c_int, [C.POINTER(snn_roi_level), C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                                C.c_void_p] + [C.c_int] * 5 + [C.POINTER(snn_params)] +
                                      [C.c_void_p] * 10 + [C.c_size_t, c_stream]),
    "snn_packed_bf16x3_elems": (C.c_size_t, [C.c_int, C.c_int]),
    "snn_packed_conv3x3_bf16x3_elems": (C.c_size_t, [C.c_int, C.c_int]),
    "snn_pack_conv3x3_weight_bf16x3": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, c_stream]),
    "snn_packed_linear_bf16x3_elems": (C.c_size_t, [C.c_int, C.c_int]),
    "snn_pack_linear_weight_bf16x3": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, c_stream]),
    "snn_check_bf16x3_split": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, c_stream]),
    "snn_pack_linear_weight_bf16x3_perm": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, c_stream]),
    "snn_det_head_forward_k": (C.c_int, [C.c_void_p] + [C.c_int] * 6 + [C.POINTER(snn_params), C.c_void_p, C.c_int] +
                               [C.c_void_p] * 9 + [C.c_size_t, c_stream]),
    "snn_det_head_forward_roialign_k": (C.c_int, [C.POINTER(snn_roi_level), C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                                  C.c_void_p] + [C.c_int] * 5 + [C.POINTER(snn_params), C.c_void_p, C.c_int] +
                                        [C.c_void_p] * 9 + [C.c_size_t, c_stream]),
    "snn_packed_linear_mx_words": (C.c_size_t, [C.c_int, C.c_int]),
    "snn_packed_conv3x3_mx_words": (C.c_size_t, [C.c_int, C.c_int]),
    "snn_pack_linear_weight_mx": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, c_stream]),
    "snn_pack_conv3x3_weight_mx": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, c_stream]),
    "snn_spike_gemm_mx": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, c_stream]),
    "snn_spike_gemm_lif_mx": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_size_t, c_stream]),
    "snn_conv3x3_lif_mx": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_size_t, c_stream]),
    "snn_spike_conv3x3_mx": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                       C.c_void_p, C.c_int, c_stream]),
    "snn_spike_gemm_bf16x3": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, c_stream]),
    "snn_spike_gemm_lif_bf16x3": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_size_t, c_stream]),
    "snn_conv3x3_lif_bf16x3": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(snn_rpn_level), C.c_int, C.c_int, C.c_int,
                                         C.c_int, C.POINTER(snn_params), C.c_void_p, C.c_void_p, C.c_size_t, c_stream]),
    "snn_spike_conv3x3_bf16x3": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(snn_rpn_level), C.c_int, C.c_int, C.c_int,
                                           C.c_int, C.c_void_p, C.c_void_p, C.c_int, c_stream]),
}

_LIB = None


class SnnHipError(RuntimeError):
    pass


class InexactWeightSplit(SnnHipError):
    """a weight tensor cannot be carried as three bf16 planes exactly (magnitudes with bits below 2^-133, values next to FLT_MAX,
    NaN / infinity): the bf16x3 kernels must not run on it (ops.check_bf16x3_split; the modules fall back to "f32_strict")"""

    def __init__(self, what: str, inexact: int, nonfinite: int):
        super().__init__("%s: %d weight(s) are not exactly hi + mid + lo in bf16, %d are non-finite" % (what, inexact, nonfinite))
        self.inexact, self.nonfinite = inexact, nonfinite


def lib_path() -> str:
    return _build.LIB_PATH


def load(build_if_missing: bool = True):
    """Load libsnnhip.so (building it in-tree if hipcc is available and it is missing/stale)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.environ.get("SNN_HIP_LIB", _build.LIB_PATH)      # override: A/B builds of the kernels
    if path == _build.LIB_PATH and build_if_missing and _build.needs_build() and os.path.exists(_build.HIPCC):
        _build.build()
    if not os.path.exists(path):
        raise SnnHipError("libsnnhip.so not found at %s (run `python -m snn_automotive_object_detection_amd.build`); "
                          "there is no CPU fallback" % path)
    lib = C.CDLL(path)
    for name, (res, args) in list(SYMBOLS.items()) + list(DEBUG_SYMBOLS.items()):
        fn = getattr(lib, name)          # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().snn_last_error()
        raise SnnHipError("%s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else "?"))


def reload_knobs():
    """the SNN_* debug knobs are read from the environment once and frozen; tests that flip them call this afterwards"""
    if _LIB is not None:
        _LIB.snn_debug_reload_knobs()
