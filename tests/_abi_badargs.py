"""Run by tests/test_abi_asan.py inside a python process that has the AddressSanitizer runtime preloaded and SNN_HIP_LIB
pointing at the ASan build of libsnnhip.so: every entry point of include/snn_hip.h is called with null / bad / boundary
arguments and - with `--deep`, on a box WITHOUT a GPU - with well-formed host-side tables, so that the host half of each
call (validation, table copies, workspace layout) runs to the point where the launch fails for lack of a device.
Exits 0 when every call returned without crashing and the bad ones were refused."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snn_automotive_object_detection_amd import _lib

deep = "--deep" in sys.argv
lib = _lib.load(build_if_missing=False)
assert os.environ.get("SNN_HIP_LIB") and "asan" in os.environ["SNN_HIP_LIB"], "expected the ASan build"
P = _lib.snn_params(0.1, -0.2, 0.0, 0.0, 0.25, 0.1, 0, 1)
refused = calls = 0


def bad(rc, what):
    global refused, calls
    calls += 1
    assert rc < 0, "%s accepted bad arguments (rc=%d)" % (what, rc)
    assert lib.snn_last_error(), what
    refused += 1


def any_rc(rc, what):
    global calls
    calls += 1
    assert isinstance(rc, int), what


FAKE = C.c_void_p(0x1000)        # a non-null "device pointer" that must never be dereferenced on the host
lv1 = (_lib.snn_rpn_level * 1)(_lib.snn_rpn_level(None, 2, 4, 4, 0))
lv_ok = (_lib.snn_rpn_level * 2)(_lib.snn_rpn_level(0x1000, 2, 8, 8, 0), _lib.snn_rpn_level(0x2000, 2, 4, 4, 0))
lv_max = (_lib.snn_rpn_level * 8)(*[_lib.snn_rpn_level(0x1000, 1, 2, 2, 0) for _ in range(8)])

# ---- size queries: total functions ----
assert lib.snn_rpn_head_workspace_bytes(None, 1, 256, 3, 8, 1) == 0
assert lib.snn_rpn_head_workspace_bytes(lv_ok, 0, 256, 3, 8, 1) == 0 and lib.snn_rpn_head_workspace_bytes(lv_ok, 9, 256, 3, 8, 1) == 0
assert lib.snn_rpn_head_workspace_bytes(lv_max, 8, 256, 3, 32, 2) > 0
assert lib.snn_det_head_workspace_bytes(0, 1, 1, 1, 1, 1, 1) == 0 and lib.snn_det_head_workspace_bytes(7, 12544, 1024, 9, 36, 12, 1) > 0
assert lib.snn_nms_workspace_bytes(1) == 8 and lib.snn_rpn_proposals_workspace_bytes(0, 5) == 0
assert lib.snn_det_postprocess_workspace_bytes(2, 0, 9) == 0 and lib.snn_det_postprocess_workspace_bytes(2, 10, 1) == 0
assert lib.snn_rpn_rates_workspace_bytes(0, 2) == 0 and lib.snn_rpn_rates_workspace_bytes(5, 2) == 5 * 2 * 2 * 32 * 8
assert lib.snn_rpn_proposals_candidates(None, 1, 3, 1000) == -1

# ---- null / bad arguments are refused before any device work ----
bad(lib.snn_pack_conv3x3_weight(None, 4, 4, None, None), "pack_conv3x3")
bad(lib.snn_pack_linear_weight(FAKE, 0, 4, FAKE, None), "pack_linear")
bad(lib.snn_pack_heads_weight(FAKE, 3, None, 12, 256, FAKE, None), "pack_heads")
bad(lib.snn_pack_conv3x3_weight_bf16x3(None, 4, 4, FAKE, None), "pack_conv3x3_bf16x3")
bad(lib.snn_pack_linear_weight_bf16x3(FAKE, 4, -1, FAKE, None), "pack_linear_bf16x3")
bad(lib.snn_pack_linear_weight_mx(FAKE, 4, 0, FAKE, None), "pack_linear_mx")
bad(lib.snn_pack_conv3x3_weight_mx(FAKE, 0, 128, FAKE, None), "pack_conv3x3_mx")
bad(lib.snn_affine_act_nchw(None, FAKE, FAKE, None, 1, 4, 16, 1, FAKE, None), "affine_act x=null")
bad(lib.snn_affine_act_nchw(FAKE, FAKE, None, None, 1, 4, 16, 1, FAKE, None), "affine_act bias=null")
bad(lib.snn_affine_act_nchw(FAKE, FAKE, FAKE, None, 1, 0, 16, 0, FAKE, None), "affine_act C=0")
bad(lib.snn_affine_act_nchw(FAKE, FAKE, FAKE, FAKE, 1 << 20, 1 << 10, 4, 1, FAKE, None), "affine_act too many planes")
bad(lib.snn_encode_nchw(None, 1, 1, 1, 1, 8, C.byref(P), None, 0, None), "encode_nchw")
bad(lib.snn_encode_nchw(FAKE, 1, 4, 2, 2, 33, C.byref(P), FAKE, 4, None), "encode_nchw T>32")
bad(lib.snn_encode_rows(None, 4, 4, 8, C.byref(P), None, 0, None), "encode_rows")
bad(lib.snn_encode_rows(FAKE, 4, 4, 0, C.byref(P), FAKE, 4, None), "encode_rows T=0")
bad(lib.snn_conv3x3_lif(None, 0, 1, 4, 4, 2, 2, 4, C.byref(P), None, None, 0, None, None, None), "conv3x3_lif")
bad(lib.snn_spike_gemm(None, 4, 32, 32, None, None, 32, None), "spike_gemm")
bad(lib.snn_spike_gemm(FAKE, 4, 32, 32, FAKE, FAKE, 16, None), "spike_gemm ldo<N")
bad(lib.snn_lif_scan(None, 4, 4, 32, 32, C.byref(P), None, 0, None, None), "lif_scan")
bad(lib.snn_li_heads(None, 0, 4, 4, 32, None, 3, 12, C.byref(P), None, None, None, None, None), "li_heads")
bad(lib.snn_li_heads(FAKE, 8, 4, 4, 32, FAKE, 3, 12, C.byref(P), FAKE, FAKE, FAKE, None, None), "li_heads one sum only")
bad(lib.snn_spike_gemm_bf16x3(None, 4, 32, 32, None, None, 32, None), "spike_gemm_bf16x3")
bad(lib.snn_spike_gemm_lif_bf16x3(FAKE, 40, 4, 32, 32, C.byref(P), FAKE, FAKE, 4, None), "spike_gemm_lif_bf16x3 T")
bad(lib.snn_spike_gemm_mx(FAKE, 4, 100, 32, FAKE, FAKE, 32, None), "spike_gemm_mx K%128")
bad(lib.snn_spike_gemm_lif_mx(FAKE, 8, 4, 100, 32, C.byref(P), FAKE, FAKE, 4, None), "spike_gemm_lif_mx K%128")
bad(lib.snn_conv3x3_lif_bf16x3(FAKE, 16, lv1, 0, 32, 32, 8, C.byref(P), FAKE, FAKE, 16, None), "conv3x3_lif_bf16x3 n_levels")
bad(lib.snn_conv3x3_lif_bf16x3(FAKE, 16, lv1, 9, 32, 32, 8, C.byref(P), FAKE, FAKE, 16, None), "conv3x3_lif_bf16x3 n_levels>8")
bad(lib.snn_spike_conv3x3_bf16x3(FAKE, 16, lv1, 1, 32, 32, 8, FAKE, FAKE, 8, None), "spike_conv3x3_bf16x3 ldo")
bad(lib.snn_conv3x3_lif_mx(FAKE, 16, lv1, 1, 100, 32, 8, C.byref(P), FAKE, FAKE, 16, None), "conv3x3_lif_mx C%128")
bad(lib.snn_spike_conv3x3_mx(FAKE, 1, lv1, 1, 128, 32, 8, FAKE, FAKE, 32, None), "spike_conv3x3_mx enc_stride")
bad(lib.snn_rpn_head_forward(None, 1, 256, 3, 8, C.byref(P), *([None] * 8), 0, None), "rpn_head_forward null")
bad(lib.snn_rpn_head_forward(lv_ok, 2, 256, 3, 8, C.byref(P), FAKE, FAKE, FAKE, FAKE, None, None, None, FAKE, 16, None),
    "rpn_head_forward small workspace")
bad(lib.snn_rpn_head_forward(lv1, 1, 256, 3, 8, C.byref(P), FAKE, FAKE, FAKE, FAKE, None, None, None, FAKE, 1 << 30, None),
    "rpn_head_forward level without features")
bad(lib.snn_rpn_head_forward_stages(lv_ok, 2, 256, 3, 8, C.byref(_lib.snn_params(0.1, -0.2, 0, 0, 0.25, 0.1, 0, 7)), FAKE, FAKE, FAKE,
                                    FAKE, None, None, None, FAKE, 1 << 30, 7, None), "rpn_head_forward precision")
bad(lib.snn_det_head_forward(None, 1, 1, 1, 1, 1, 40, C.byref(P), *([None] * 10), 0, None), "det_head_forward null")
bad(lib.snn_det_head_forward(FAKE, 4, 64, 32, 3, 12, 8, C.byref(P), FAKE, FAKE, FAKE, FAKE, FAKE, None, None, None, None, FAKE, 8, None),
    "det_head_forward small workspace")
roi_lv = (_lib.snn_roi_level * 4)(*[_lib.snn_roi_level(0x1000, 8, 8, 0.25, 0) for _ in range(4)])
bad(lib.snn_roi_align_encode(roi_lv, 5, 8, FAKE, FAKE, FAKE, 4, 8, C.byref(P), FAKE, 13, None, None), "roi_align_encode n_levels>4")
bad(lib.snn_roi_align_encode(roi_lv, 4, 8, None, FAKE, FAKE, 4, 8, C.byref(P), FAKE, 13, None, None), "roi_align_encode null rois")
bad(lib.snn_det_head_forward_roialign(roi_lv, 4, 8, FAKE, FAKE, FAKE, 4, 32, 3, 12, 8, C.byref(P), FAKE, FAKE, FAKE, FAKE, FAKE,
                                      None, None, None, None, FAKE, 8, None), "det_head_forward_roialign small workspace")
bad(lib.snn_nms_sorted(None, None, 4, 0.5, 4, None, None, None, 0, None), "nms_sorted")
bad(lib.snn_nms_sorted(FAKE, None, 20000, 0.5, 4, FAKE, FAKE, FAKE, 1 << 30, None), "nms_sorted n too large")
bad(lib.snn_nms_sorted(FAKE, None, 100, 0.5, 4, FAKE, FAKE, FAKE, 8, None), "nms_sorted small workspace")
post_lv = (_lib.snn_rpn_post_level * 2)()
for l in range(2):
    post_lv[l].logits, post_lv[l].deltas, post_lv[l].H, post_lv[l].W = 0x1000, 0x2000, 8 >> l, 8 >> l
    post_lv[l].stride_h = post_lv[l].stride_w = float(4 << l)
hw = (C.c_float * 4)(32.0, 32.0, 30.0, 31.0)
assert lib.snn_rpn_proposals_candidates(post_lv, 2, 3, 100) == 100 + 48
bad(lib.snn_rpn_proposals(post_lv, 2, 2, 3, None, 100, 100, 0.7, 0.0, 1e-3, FAKE, FAKE, FAKE, None, None, FAKE, 1 << 30, None), "rpn_proposals null image sizes")
bad(lib.snn_rpn_proposals(post_lv, 2, 2, 17, hw, 100, 100, 0.7, 0.0, 1e-3, FAKE, FAKE, FAKE, None, None, FAKE, 1 << 30, None), "rpn_proposals A>16")
bad(lib.snn_rpn_proposals(post_lv, 2, 65, 3, hw, 100, 100, 0.7, 0.0, 1e-3, FAKE, FAKE, FAKE, None, None, FAKE, 1 << 30, None), "rpn_proposals N>64")
bad(lib.snn_rpn_proposals(post_lv, 2, 2, 3, hw, 100, 100, 0.7, 0.0, 1e-3, FAKE, FAKE, FAKE, None, None, FAKE, 64, None), "rpn_proposals small workspace")
rpi = (C.c_int * 2)(5, 3)
bw = (C.c_float * 4)(10.0, 10.0, 5.0, 5.0)
bad(lib.snn_det_postprocess(None, FAKE, FAKE, rpi, 2, 9, hw, bw, 0.4, 0.5, 100, 1e-2, *([FAKE] * 6), 200, FAKE, 1 << 30, None), "det_postprocess null")
bad(lib.snn_det_postprocess(FAKE, FAKE, FAKE, (C.c_int * 2)(5, -1), 2, 9, hw, bw, 0.4, 0.5, 100, 1e-2, *([FAKE] * 6), 200, FAKE, 1 << 30, None),
    "det_postprocess negative RoI count")
bad(lib.snn_det_postprocess(FAKE, FAKE, FAKE, rpi, 2, 9, hw, bw, 0.4, 0.5, 100, 1e-2, *([FAKE] * 6), 50, FAKE, 1 << 30, None), "det_postprocess out_cap")
bad(lib.snn_det_postprocess(FAKE, FAKE, FAKE, rpi, 2, 9, hw, bw, 0.4, 0.5, 100, 1e-2, *([FAKE] * 6), 200, FAKE, 16, None), "det_postprocess small workspace")
bad(lib.snn_det_postprocess(FAKE, FAKE, FAKE, (C.c_int * 2)(5000, 3), 2, 9, hw, bw, 0.4, 0.5, 100, 1e-2, *([FAKE] * 6), 6000, FAKE, 1 << 40, None),
    "det_postprocess too many candidates")
bad(lib.snn_det_exchange_payload(None, FAKE, 2, 10, 9, 5, FAKE, FAKE, None), "det_exchange_payload null")
bad(lib.snn_det_exchange_payload(FAKE, FAKE, 2, 5000, 9, 5, FAKE, FAKE, None), "det_exchange_payload too many RoIs")
bad(lib.snn_rpn_rates(lv_ok, 2, 256, 3, 8, None, FAKE, FAKE, FAKE, FAKE, 1 << 20, None), "rpn_rates null")
bad(lib.snn_rpn_rates(lv_ok, 2, 256, 3, 8, FAKE, FAKE, FAKE, FAKE, FAKE, 8, None), "rpn_rates small workspace")
bad(lib.snn_det_rates(4, 64, 32, 3, 12, 8, 0, None, FAKE, FAKE, FAKE, FAKE, None), "det_rates null")
bad(lib.snn_det_rates(4, 64, 32, 3, 12, 0, 0, FAKE, FAKE, FAKE, FAKE, FAKE, None), "det_rates T=0")
o12 = (C.c_int32 * 12)()
bad(lib.snn_debug_tile_shape(1, 1000, 256, 256, 8, 0, 0, None), "tile_shape null out")
bad(lib.snn_debug_tile_shape(1, 0, 256, 256, 8, 0, 0, o12), "tile_shape no units")
bad(lib.snn_debug_tile_shape(0, 2000, 12544, 1024, 33, 0, 6, o12), "tile_shape T > 32")
# host-only calls.  Default knobs: the structured-sparse plans (csrc/snn_sparse.h) - conv T = 8: 7 live steps = 2 dense + 5 sparse period
# planes on tiles of 64 positions; fc6 T = 12: 10 planes on tiles of 32 RoIs, 4 x 2 wave grid; spike-rate mode one step more; T = 24: 22 planes, 16 RoIs
assert lib.snn_debug_tile_shape(1, 196416, 256, 256, 8, 0, 0, o12) == 0 and list(o12[2:11]) == [448, 64, 7, 4 * 3069, 4, 1, 1, 2, 5], list(o12)
assert lib.snn_debug_tile_shape(0, 2000, 12544, 1024, 12, 0, 6, o12) == 0 and list(o12[2:11]) == [320, 32, 10, 63 * 16, 16, 2, 1, 2, 8], list(o12)
assert lib.snn_debug_tile_shape(0, 2000, 12544, 1024, 12, 1, 6, o12) == 0 and o12[4] == 11 and o12[8] == 1, list(o12)                  # spike-rate mode: fc6 0 .. T-2
assert lib.snn_debug_tile_shape(0, 2000, 12544, 1024, 24, 0, 6, o12) == 0 and list(o12[2:5]) == [352, 16, 22] and o12[8] == 1 and o12[11] == 22, list(o12)
assert lib.snn_debug_tile_shape(0, 2000, 12544, 1024, 24, 1, 6, o12) == 0 and o12[4] == 23 and o12[8] == 1, list(o12)                  # config[4]: T_det = 24, rates on
for T in range(6, 27):                                                                # every T_det of the reference's range has a sparse plan
    assert lib.snn_debug_tile_shape(0, 2000, 12544, 1024, T, 0, 6, o12) == 0 and o12[8] == 1 and o12[4] == T - 2 and o12[11] == o12[4] * (o12[3] // 16), (T, list(o12))
assert lib.snn_debug_tile_shape(1, 196416, 256, 256, 4, 0, 0, o12) == 0 and o12[8] == 0 and o12[4] == 3                               # T_rpn = 4: the dense tile
assert lib.snn_debug_tile_shape(1, 196416, 192, 192, 8, 0, 0, o12) == 0 and o12[8] == 0                                                # 3 column blocks: no XCD grouping
assert lib.snn_debug_tile_shape(0, 2000, 1024, 1024, 12, 0, 7, o12) == 0 and o12[8] == 0 and o12[4] == 10 and o12[3] * o12[4] <= o12[2]  # fc7: dense tile
assert lib.snn_debug_tile_shape(0, 2000, 1024, 1024, 1, 0, 7, o12) == 0 and o12[4] == 1
os.environ["SNN_SPARSE"] = "0"
lib.snn_debug_reload_knobs()
assert lib.snn_debug_tile_shape(1, 196416, 256, 256, 8, 0, 0, o12) == 0 and o12[4] == 7 and o12[2] == 512 and o12[3] == 73 and o12[8] == 0   # the round-3 dense tile
del os.environ["SNN_SPARSE"]
lib.snn_debug_reload_knobs()

# ---- well-formed host tables, no device: the whole host half runs, the first launch (or attribute call) fails with -3 ----
if deep:
    big = 1 << 34
    for prec in (0, 1, 2):
        Pp = _lib.snn_params(0.1, -0.2, 0.0, 0.0, 0.25, 0.1, 0, prec)
        lvC = (_lib.snn_rpn_level * 8)(*[_lib.snn_rpn_level(0x1000, 2, 3 + l, 5 + l, 0) for l in range(8)])
        any_rc(lib.snn_rpn_head_forward(lvC, 8, 128, 3, 8, C.byref(Pp), FAKE, FAKE, FAKE, FAKE, FAKE, FAKE, FAKE, FAKE, big, None), "rpn_head deep")
        any_rc(lib.snn_det_head_forward(FAKE, 37, 49 * 128, 128, 9, 36, 12, C.byref(Pp), FAKE, FAKE, FAKE, FAKE, FAKE, FAKE, FAKE, FAKE, FAKE,
                                        FAKE, big, None), "det_head deep")
        any_rc(lib.snn_det_head_forward_roialign(roi_lv, 4, 128, FAKE, FAKE, FAKE, 5, 128, 9, 36, 12, C.byref(Pp), FAKE, FAKE, FAKE, FAKE,
                                                 FAKE, None, None, None, None, FAKE, big, None), "det_head_roialign deep")
    full_lv = (_lib.snn_rpn_post_level * 8)()
    for l in range(8):
        full_lv[l].logits, full_lv[l].deltas, full_lv[l].H, full_lv[l].W = 0x1000, 0x2000, 40 >> (l // 2), 40 >> (l // 2)
        full_lv[l].stride_h = full_lv[l].stride_w = 4.0
        for a_ in range(16):
            for q in range(4):
                full_lv[l].base_anchors[a_][q] = float(a_ + q)
    hw64 = (C.c_float * 128)(*([32.0] * 128))
    any_rc(lib.snn_rpn_proposals(full_lv, 8, 64, 16, hw64, 1000, 1000, 0.7, 0.0, 1e-3, FAKE, FAKE, FAKE, FAKE, FAKE, FAKE, big, None), "rpn_proposals deep")
    rpi64 = (C.c_int * 64)(*([30] * 64))
    any_rc(lib.snn_det_postprocess(FAKE, FAKE, FAKE, rpi64, 64, 9, hw64, bw, 0.4, 0.5, 100, 1e-2, *([FAKE] * 6), 200, FAKE, big, None), "det_postprocess deep")
    any_rc(lib.snn_det_postprocess(FAKE, FAKE, FAKE, (C.c_int * 2)(0, 0), 2, 9, hw, bw, 0.4, 0.5, 100, 1e-2, *([FAKE] * 6), 200, FAKE, big, None), "det_postprocess empty")
    any_rc(lib.snn_rpn_rates(lv_max, 8, 256, 3, 8, FAKE, FAKE, FAKE, FAKE, FAKE, 1 << 20, None), "rpn_rates deep")
    any_rc(lib.snn_det_rates(100, 12544, 1024, 9, 36, 12, 0, FAKE, FAKE, FAKE, FAKE, FAKE, None), "det_rates deep")
    any_rc(lib.snn_nms_sorted(FAKE, FAKE, 9000, 0.5, 100, FAKE, FAKE, FAKE, big, None), "nms_sorted deep")
    any_rc(lib.snn_li_heads(FAKE, 64, 8, 100, 256, FAKE, 3, 12, C.byref(P), FAKE, FAKE, None, None, None), "li_heads deep")
    any_rc(lib.snn_det_exchange_payload(FAKE, FAKE, 2, 1000, 9, 100, FAKE, FAKE, None), "det_exchange_payload deep")
print("ABI_BADARGS_OK calls=%d refused=%d deep=%s" % (calls, refused, deep), flush=True)
