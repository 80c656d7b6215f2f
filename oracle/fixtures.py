"""ORACLE tooling — fixture specs shared by oracle/make_golden.py (which writes the expected
outputs by running the shimmed reference) and by tests/ (which regenerate the SAME inputs and
weights from oracle/portable_rng.py and compare against the committed outputs).

A fixture = a spec (below; plain data) + tests/golden/<name>.npz (expected outputs, bit-packed
per-step spikes).  Inputs/weights are deterministic functions of the spec.
"""
import os

import numpy as np
import torch

from . import portable_rng as PR

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

# name -> spec.   RPN: C, A, T, shapes, seed, w_std (reference init is N(0, 0.01²), rpn.py:78-82;
# the fixtures use a bell-shaped stand-in with the same std unless noted), feat_std/feat_mean.
RPN_SPECS = {
    "rpn_c16_T8":      dict(C=16, A=3, T=8, shapes=[(2, 16, 12, 24), (2, 16, 6, 12)], seed=101, w_std=0.05),
    "rpn_c256_T8":     dict(C=256, A=3, T=8, shapes=[(2, 256, 12, 24)], seed=102, w_std=0.01),
    "rpn_c256_T8_odd": dict(C=256, A=3, T=8, shapes=[(1, 256, 13, 22), (2, 256, 5, 7)], seed=103, w_std=0.01),
    "rpn_c256_T4":     dict(C=256, A=3, T=4, shapes=[(1, 256, 6, 10)], seed=104, w_std=0.01, feat_std=2.0),
    "rpn_c256_T12":    dict(C=256, A=3, T=12, shapes=[(1, 256, 6, 10)], seed=112, w_std=0.01),
    "rpn_c256_T16":    dict(C=256, A=3, T=16, shapes=[(1, 256, 6, 10)], seed=116, w_std=0.01),
    "rpn_c256_T24":    dict(C=256, A=3, T=24, shapes=[(1, 256, 6, 10)], seed=124, w_std=0.01),
    "rpn_c64_A5_T8":   dict(C=64, A=5, T=8, shapes=[(3, 64, 9, 17), (3, 64, 1, 1)], seed=130, w_std=0.03),
    # a pyramid with the 5 Cityscapes level aspect ratios at 1/8 scale (24x48 ... 2x3)
    "rpn_c256_T8_pyr": dict(C=256, A=3, T=8,
                            shapes=[(2, 256, 24, 48), (2, 256, 12, 24), (2, 256, 6, 12), (2, 256, 3, 6), (2, 256, 2, 3)],
                            seed=140, w_std=0.01),
}

# DET: C (D = C*49), Hd, K, T, R, seed.  Default nn.Linear init is U(+-1/sqrt(fan_in))
# (faster_rcnn.py:447-467 use it unchanged); the fixtures use exactly that distribution.
DET_SPECS = {
    "det_K9_T12":         dict(C=256, Hd=1024, K=9, T=12, R=16, seed=201),
    "det_K11_T8_R37":     dict(C=256, Hd=1024, K=11, T=8, R=37, seed=202),
    "det_small_T16":      dict(C=8, Hd=64, K=5, T=16, R=9, seed=203),
    "det_K9_T24_onebbox": dict(C=256, Hd=1024, K=9, T=24, R=5, seed=204, only_one_bbox=True),
    "det_K9_T12_R130":    dict(C=256, Hd=1024, K=9, T=12, R=130, seed=205),
}


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def rpn_inputs(spec):
    C, A, s = spec["C"], spec["A"], spec["seed"]
    w_std = spec.get("w_std", 0.01)
    w_shared = _t(PR.normalish((C, C, 3, 3), s * 10 + 1, std=w_std))
    w_cls = _t(PR.normalish((A, C, 1, 1), s * 10 + 2, std=w_std))
    w_bbox = _t(PR.normalish((4 * A, C, 1, 1), s * 10 + 3, std=w_std))
    feats = [_t(PR.normalish(shape, s * 10 + 4 + l, std=spec.get("feat_std", 1.0),
                             mean=spec.get("feat_mean", 0.0)))
             for l, shape in enumerate(spec["shapes"])]
    return feats, w_shared, w_cls, w_bbox


def det_inputs(spec):
    C, Hd, K, R, s = spec["C"], spec["Hd"], spec["K"], spec["R"], spec["seed"]
    D = C * 49
    K4 = 4 if spec.get("only_one_bbox", False) else 4 * K
    b6, b7 = 1.0 / D ** 0.5, 1.0 / Hd ** 0.5
    w6 = _t(PR.uniform((Hd, D), s * 10 + 1, -b6, b6))
    w7 = _t(PR.uniform((Hd, Hd), s * 10 + 2, -b7, b7))
    w_cls = _t(PR.uniform((K, Hd), s * 10 + 3, -b7, b7))
    w_bbox = _t(PR.uniform((K4, Hd), s * 10 + 4, -b7, b7))
    x = _t(PR.normalish((R, C, 7, 7), s * 10 + 5, std=spec.get("feat_std", 1.0)))
    return x, w6, w7, w_cls, w_bbox


# ---------------------------------------------------------------------------------------------
# energy report: tests/golden/energy_*.npz hold what the reference's own block train.py:472-515 computes
# from these rate lists (oracle/make_golden.py execs it); the lists have the layout the spike-rate forwards return (rpn.py:177-200:
# 3 entries per level, [N, 2] = (rate, FLOPs); faster_rcnn.py:594-618: 4 entries [R, 2])
# ---------------------------------------------------------------------------------------------
ENERGY_SPECS = {
    "energy_city_T8_T12":  dict(seed=501, images=6, rois=2000, levels=[(192, 384), (96, 192), (48, 96), (24, 48), (12, 24)], C=256, A=3, K=9, T_rpn=8, T_det=12),
    "energy_bdd_T16_T24":  dict(seed=502, images=4, rois=1500, levels=[(192, 344), (96, 172), (48, 86), (24, 43), (12, 22)], C=256, A=3, K=11, T_rpn=16, T_det=24),
    "energy_small_T4_T6":  dict(seed=503, images=1, rois=7, levels=[(8, 16), (4, 8)], C=16, A=3, K=5, T_rpn=4, T_det=6),
}


def energy_rates(spec):
    """{list position: [*, 2] float32 (rate, FLOPs)} with the FLOP constants of rpn.py:177-188 / faster_rcnn.py:594-603 (labels swapped
    for obj / bbox, as the reference has them) and seeded rates in (0, 0.3)"""
    s, n, C, A, K = spec["seed"], spec["images"], spec["C"], spec["A"], spec["K"]
    rates = {}
    for l, (h, w) in enumerate(spec["levels"]):
        for j, fl in enumerate((9 * h * w * C * C, h * w * C * A * 4, h * w * C * A)):
            r = PR.uniform((n,), s * 100 + 3 * l + j, 0.0, 0.3)
            rates[3 * l + j] = _t(np.stack([r, np.full((n,), float(fl), dtype=np.float32)], 1).astype(np.float32))
    base = 3 * len(spec["levels"])
    D, Hd = C * 49, 1024
    for j, fl in enumerate((D * Hd, Hd * Hd, Hd * K, Hd * K * 4)):
        r = PR.uniform((spec["rois"],), s * 100 + 50 + j, 0.0, 0.3)
        rates[base + j] = _t(np.stack([r, np.full((spec["rois"],), float(fl), dtype=np.float32)], 1).astype(np.float32))
    return rates


def load_expected(name):
    with np.load(os.path.join(GOLDEN_DIR, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


def unpack_spikes(packed: np.ndarray, shape) -> np.ndarray:
    n = int(np.prod(shape))
    return np.unpackbits(packed)[:n].reshape(tuple(int(s) for s in shape)).astype(np.float32)


# ---------------------------------------------------------------------------------------------
# post-processing fixtures: tests/golden/post_*.npz hold what the reference's own
# RegionProposalNetwork.forward (rpn.py:563-703) / RoIHeadsSNN.postprocess_detections (roi_heads.py:1075-1176)
# return for these inputs (oracle/make_golden.py)
# ---------------------------------------------------------------------------------------------
ANCHOR_SIZES = ((32,), (64,), (128,), (256,), (512,))            # faster_rcnn.py:31-34
ASPECT_RATIOS = ((0.5, 1.0, 2.0),) * 5

# canvas = padded batch tensor (H, W); image_sizes = per-image size after the transform; grids = FPN maps; A = 3.
# model.py:50-59: pre/post_nms_top_n 1000, nms 0.7, score_thresh 0.0
RPN_POST_SPECS = {
    # 2 images of different size on a 192x320 canvas; level 0 has 48*80*3 = 11520 anchors (> 1000: the per-level top-k bites)
    "post_rpn_base":   dict(canvas=(192, 320), image_sizes=[(192, 320), (170, 301)], grids=[(48, 80), (24, 40), (12, 20), (6, 10), (3, 5)],
                            seed=301, logit_std=2.0, delta_std=0.5, pre=1000, post=1000, nms=0.7, score_thresh=0.0),
    # proposals mostly unmoved anchors (small deltas, like a random-init head): heavy NMS suppression between neighbours
    "post_rpn_tight":  dict(canvas=(192, 320), image_sizes=[(192, 320)], grids=[(48, 80), (24, 40), (12, 20), (6, 10), (3, 5)],
                            seed=302, logit_std=0.05, delta_std=0.02, pre=1000, post=1000, nms=0.7, score_thresh=0.0),
    # truncation after NMS, a score threshold that removes candidates (>= keeps logits that are exactly 0 at thresh 0.5),
    # and boxes collapsed below min_size by very negative size deltas
    "post_rpn_edges":  dict(canvas=(96, 160), image_sizes=[(96, 160), (90, 150), (64, 100)], grids=[(24, 40), (12, 20), (6, 10), (3, 5), (2, 3)],
                            seed=303, logit_std=1.5, delta_std=0.7, pre=300, post=50, nms=0.7, score_thresh=0.5,
                            zero_logit_every=7, collapse_every=11),
    # an image that ends up with no proposal at all (threshold above every sigmoid)
    "post_rpn_empty":  dict(canvas=(64, 96), image_sizes=[(64, 96), (64, 96)], grids=[(16, 24), (8, 12), (4, 6), (2, 3), (1, 2)],
                            seed=304, logit_std=1.0, delta_std=0.3, pre=1000, post=1000, nms=0.7, score_thresh=0.9999,
                            boost_image=1),
    # quantised logits: many exactly tied scores (order inside a tie is unspecified in the reference; tests compare tie-aware)
    "post_rpn_ties":   dict(canvas=(96, 160), image_sizes=[(96, 160)], grids=[(24, 40), (12, 20), (6, 10), (3, 5), (2, 3)],
                            seed=305, logit_std=1.0, delta_std=0.4, pre=200, post=100, nms=0.7, score_thresh=0.0, quant=0.25),
}

# model.py:98-106: score_thresh 0.4, nms 0.5, 100 detections per image, box weights (10,10,5,5)
DET_POST_SPECS = {
    "post_det_base":    dict(K=9, rois=[300, 200], image_shapes=[(768, 1536), (700, 1400)], seed=401, logit_std=2.5, delta_std=1.0),
    # every RoI background: only the background list survives
    "post_det_bgonly":  dict(K=9, rois=[120], image_shapes=[(768, 1536)], seed=402, logit_std=0.3, delta_std=0.5, bg_bias=6.0),
    # > 100 foreground candidates per image after NMS (truncation to detections_per_img), K = 11 (BDD)
    "post_det_many":    dict(K=11, rois=[400, 350], image_shapes=[(768, 1376), (768, 1376)], seed=403, logit_std=4.0, delta_std=0.3,
                             fg_bias=2.0, spread=True),
    # an image without RoIs, one with a single RoI, degenerate (collapsed) boxes
    "post_det_ragged":  dict(K=9, rois=[0, 1, 57], image_shapes=[(768, 1536), (768, 1536), (400, 900)], seed=404, logit_std=2.5,
                             delta_std=1.0, collapse_every=5),
    # clustered RoIs (proposals piled on a few objects, as after a trained RPN): dense NMS interaction per class
    "post_det_cluster": dict(K=9, rois=[250, 250], image_shapes=[(768, 1536), (768, 1536)], seed=405, logit_std=3.0, delta_std=0.2,
                             clusters=6),
}


# RoI pairs whose IoU sits within ~1e-5 of the NMS threshold: torchvision's coordinate trick (<= 4000 coordinates: every class
# shifted by label * (max coordinate + 1) before ONE nms) decides some of them differently from the raw coordinates
DET_POST_SPECS["post_det_trick"] = dict(K=9, rois=[96, 96], image_shapes=[(768, 1536), (768, 1536)], seed=406, pairs=True)


def _det_trick_inputs(spec):
    K, s = spec["K"], spec["seed"]
    R = sum(spec["rois"])
    logits = np.zeros((R, K), dtype=np.float32)
    reg = np.zeros((R, 4 * K), dtype=np.float32)                 # zero deltas: decoded box = proposal (up to an ulp)
    props, base = [], 0
    for i, (r, (h, w)) in enumerate(zip(spec["rois"], spec["image_shapes"])):
        n_pairs = r // 2
        u = PR.uniform((n_pairs, 6), s * 100 + 10 + i, 0.0, 1.0)
        cls = 1 + (np.arange(n_pairs) % (K - 1))                 # class of the pair
        cell = np.arange(n_pairs) // (K - 1)                     # pairs of one class sit in different cells (no interaction)
        cx0 = 20 + (cell % 8) * 185.0
        cy0 = 20 + (cell // 8) * 140.0
        bw = (40 + 100 * u[:, 0]).astype(np.float32)
        bh = (40 + 70 * u[:, 1]).astype(np.float32)
        x1 = (cx0 + 10 * u[:, 2]).astype(np.float32)
        y1 = (cy0 + 10 * u[:, 3]).astype(np.float32)
        dx = (bw / 3 * (1.0 + (u[:, 4] - 0.5) * 4e-5)).astype(np.float32)      # IoU of the pair = 0.5 +- ~1e-5
        a = np.stack([x1, y1, x1 + bw, y1 + bh], 1).astype(np.float32)
        b = a.copy()
        b[:, 0] += dx
        b[:, 2] += dx
        pr = np.empty((2 * n_pairs, 4), dtype=np.float32)
        pr[0::2], pr[1::2] = a, b
        props.append(_t(pr))
        rows = base + np.arange(n_pairs) * 2
        logits[rows, cls] = 8.5                                  # the first box of a pair scores higher: it is walked first
        logits[rows + 1, cls] = 8.0
        base += r
    return _t(logits), _t(reg), props


def rpn_post_inputs(spec):
    """-> (objectness [N,A,H,W] per level, deltas [N,4A,H,W] per level) as the head returns them (NCHW)"""
    N, A, s = len(spec["image_sizes"]), 3, spec["seed"]
    obj, dl = [], []
    for l, (h, w) in enumerate(spec["grids"]):
        o = PR.normalish((N, A, h, w), s * 100 + 2 * l, std=spec["logit_std"])
        d = PR.normalish((N, 4 * A, h, w), s * 100 + 2 * l + 1, std=spec["delta_std"])
        if spec.get("quant"):
            o = (np.round(o / spec["quant"]) * spec["quant"]).astype(np.float32)
        flat = o.reshape(-1)
        if spec.get("zero_logit_every"):
            flat[:: spec["zero_logit_every"]] = 0.0                      # sigmoid = 0.5 exactly: kept by `>= 0.5`
        if spec.get("collapse_every"):
            d4 = d.reshape(N, A, 4, h, w)
            m = (np.arange(N * A * h * w) % spec["collapse_every"] == 0).reshape(N, A, h, w)
            d4[:, :, 2][m] = -40.0                                       # width exp(-40) * w < min_size
            d = d4.reshape(N, 4 * A, h, w)
        if spec.get("boost_image") is not None:
            o[spec["boost_image"]] += 12.0                               # sigmoid > 0.9999 for this image only
        t = spec["score_thresh"]
        if 0.0 < t < 1.0:
            # keep the candidates off the knife edge of the score threshold: sigmoid implementations differ in the last ulp, so
            # a logit within ~1e-3 of logit(t) could pass on one side and not on the other (exact hits, sigmoid(0) = 0.5, stay)
            lt = np.float32(np.log(t / (1.0 - t)))
            near = (np.abs(o - lt) < 0.25) & (o != lt)
            o[near] = o[near] + np.where(o[near] >= lt, np.float32(0.25), np.float32(-0.25))
        obj.append(_t(o))
        dl.append(_t(d))
    return obj, dl


def det_post_inputs(spec):
    """-> (class_logits [R,K], box_regression [R,4K], proposals: list of [R_i,4])"""
    if spec.get("pairs"):
        return _det_trick_inputs(spec)
    K, s = spec["K"], spec["seed"]
    R = sum(spec["rois"])
    logits = PR.normalish((R, K), s * 100 + 1, std=spec["logit_std"])
    if spec.get("bg_bias"):
        logits[:, 0] += spec["bg_bias"]
    if spec.get("fg_bias"):
        logits[:, 1:] += spec["fg_bias"] * (PR.uniform((R, K - 1), s * 100 + 5, 0.0, 1.0) > 0.8)
    reg = PR.normalish((R, 4 * K), s * 100 + 2, std=spec["delta_std"])
    props, base = [], 0
    for i, (r, (h, w)) in enumerate(zip(spec["rois"], spec["image_shapes"])):
        u = PR.uniform((r, 4), s * 100 + 10 + i, 0.0, 1.0)
        if spec.get("clusters"):
            c = PR.uniform((spec["clusters"], 4), s * 100 + 50 + i, 0.1, 0.9)
            which = (np.arange(r) % spec["clusters"])
            cx = c[which, 0] * w + (u[:, 0] - 0.5) * 12; cy = c[which, 1] * h + (u[:, 1] - 0.5) * 12
            bw = (40 + 200 * c[which, 2]) * (0.9 + 0.2 * u[:, 2]); bh = (40 + 150 * c[which, 3]) * (0.9 + 0.2 * u[:, 3])
        else:
            cx, cy = u[:, 0] * w, u[:, 1] * h
            bw = np.exp(np.log(16.0) + u[:, 2] * np.log(32.0)); bh = np.exp(np.log(16.0) + u[:, 3] * np.log(24.0))
        b = np.stack([cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2], 1).astype(np.float32)
        b[:, 0::2] = np.clip(b[:, 0::2], 0, w); b[:, 1::2] = np.clip(b[:, 1::2], 0, h)
        props.append(_t(b))
        if spec.get("collapse_every") and r:
            rows = np.arange(base, base + r)[:: spec["collapse_every"]]
            reg.reshape(R, K, 4)[rows, :, 2] = -300.0                    # dw / 5 = -60: width below min_size for every class
        base += r
    if spec.get("spread"):                                              # keep boxes apart so that > 100 survive NMS
        reg *= np.float32(0.2)
    return _t(logits.astype(np.float32)), _t(reg.astype(np.float32)), props


# ---------------------------------------------------------------------------------------------
# box-pooling fixtures.  torchvision is absent from /root/reference and not installed here, so
# these expected values come from the restatement (oracle/roi_align_oracle.py, pinned by closed-form known answers in
# tests/test_roi_align_oracle.py), NOT from a run of the reference - the header of that module says "parity unpinned".
# The pooled features then go through the (reference-pinned) detector-head oracle: expected cls / bbox of the fused
# RoIAlign + encoder + head path.
# ---------------------------------------------------------------------------------------------
ROI_SPECS = {
    # 4 FPN levels of a 192x320 image at strides 4..32 (+ the unused 'pool' level), 2 images, boxes over all four levels,
    # three corner cases per image (partly outside, beyond the far edge, degenerate)
    "roialign_c8": dict(C=8, sizes=[(48, 80), (24, 40), (12, 20), (6, 10)], pool=(3, 5), image_shapes=[(192, 320), (180, 300)],
                        rois=[60, 45], seed=501, Hd=64, K=5, T=12),
}


def roi_inputs(spec):
    """-> (features {name: [N,C,H,W]}, boxes per image [k,4], image_shapes)"""
    C, s = spec["C"], spec["seed"]
    n_img = len(spec["image_shapes"])
    feats = {str(i): _t(PR.normalish((n_img, C, h, w), s * 10 + i, std=1.0)) for i, (h, w) in enumerate(spec["sizes"])}
    feats["pool"] = _t(PR.normalish((n_img, C) + tuple(spec["pool"]), s * 10 + 9, std=1.0))
    boxes = []
    for n, (k, (ih, iw)) in enumerate(zip(spec["rois"], spec["image_shapes"])):
        u = PR.uniform((k, 4), s * 10 + 20 + n, 0.0, 1.0).astype(np.float64)
        xy = u[:, :2] * np.array([iw * 0.95, ih * 0.95])
        wh = np.exp(u[:, 2:] * 5.0 + 1.0)                         # 2.7 .. 400 px: all four levels
        b = np.concatenate([xy, xy + wh], 1).astype(np.float32)
        b[0] = [-20.0, -10.0, 30.0, 25.0]                         # partly outside
        b[1] = [iw - 5.0, ih - 5.0, iw + 80.0, ih + 110.0]        # beyond the far edge
        b[2] = [100.0, 100.0, 100.2, 100.1]                       # degenerate: clamps to one feature pixel
        boxes.append(_t(b))
    return feats, boxes, [tuple(sh) for sh in spec["image_shapes"]]


def roi_head_weights(spec):
    """weights of the small detector head behind the pooled features (same distribution as det_inputs)"""
    C, Hd, K, s = spec["C"], spec["Hd"], spec["K"], spec["seed"]
    D = C * 49
    b6, b7 = 1.0 / D ** 0.5, 1.0 / Hd ** 0.5
    return (_t(PR.uniform((Hd, D), s * 10 + 31, -b6, b6)), _t(PR.uniform((Hd, Hd), s * 10 + 32, -b7, b7)),
            _t(PR.uniform((K, Hd), s * 10 + 33, -b7, b7)), _t(PR.uniform((4 * K, Hd), s * 10 + 34, -b7, b7)))
