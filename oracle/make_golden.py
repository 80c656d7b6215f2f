#!/usr/bin/env python3
"""ORACLE tooling — generates tests/golden/*.npz.  Runs ONLY in the build container (needs
/root/reference); nothing here runs on the GPU box.

What it does:
  * pre-populates sys.modules with inert ``torchvision`` placeholders and with ``norse`` modules
    that resolve to oracle/norse_restated.py (Norse 0.0.7 is not vendored in the reference and not
    installed here),
  * imports the reference's own ``rpn`` and ``faster_rcnn`` modules from /root/reference,
  * constructs ``RPNHeadSNN`` / ``FastRCNNPredictorSNNFull`` and executes THEIR ``forward`` bodies
    unmodified (the spike-rate variants live in string literals at rpn.py:126-200 and
    faster_rcnn.py:520-618 and are exec-ed verbatim from the reference text at run time),
  * stores inputs, weights and outputs as small fixtures (tensors, not seeds),
  * cross-checks the oracle restatement (oracle/snn_oracle.py) against those outputs bit-for-bit.

The fixtures are DATA (inputs + expected outputs); no reference source text is stored.

Usage:  python oracle/make_golden.py [--out tests/golden]
"""
import argparse
import contextlib
import io
import os
import sys
import textwrap
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
REF = "/root/reference"

from oracle import norse_restated as NR          # noqa: E402
from oracle import snn_oracle as OR              # noqa: E402
from oracle import fixtures as FX                # noqa: E402
from oracle import post_oracle as PO             # noqa: E402
from oracle import torchvision_restated as TV    # noqa: E402


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__file__ = os.path.join(HERE, "_shim_" + name.replace(".", "_") + ".py")
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    parent, _, child = name.rpartition(".")
    if parent:
        setattr(sys.modules[parent], child, m)
    return m


class _Inert:
    """placeholder for torchvision classes only named at import / class-creation time"""
    def __init__(self, *a, **k):
        pass


def install_shims():
    # ---- norse (4 symbols; reference imports at rpn.py:16-19, faster_rcnn.py:24-27) ----
    _mod("norse")
    _mod("norse.torch", LIFParameters=NR.LIFParameters, LICell=NR.LICell, LIFCell=NR.LIFCell)
    _mod("norse.torch.module")
    _mod("norse.torch.module.lif", LIFCell=NR.LIFCell)
    _mod("norse.torch.functional")
    _mod("norse.torch.functional.lif", lif_current_encoder=NR.lif_current_encoder,
         LIFParameters=NR.LIFParameters)
    # ---- torchvision placeholders (names touched at import time only) ----
    # (the box ops, BoxCoder, AnchorGenerator and ImageList resolve to oracle/torchvision_restated.py so that the reference's
    # own RegionProposalNetwork.forward / RoIHeadsSNN.postprocess_detections bodies can run: rows f2 / f3)
    _mod("torchvision", _is_tracing=TV._is_tracing)
    _mod("torchvision.ops", MultiScaleRoIAlign=_Inert, roi_align=lambda *a, **k: None)
    _mod("torchvision.ops.boxes", clip_boxes_to_image=TV.clip_boxes_to_image, remove_small_boxes=TV.remove_small_boxes,
         batched_nms=TV.batched_nms, nms=TV.nms, box_iou=TV.box_iou)
    _mod("torchvision.ops.misc", FrozenBatchNorm2d=_Inert, Conv2dNormActivation=_Inert)
    _mod("torchvision.models")
    _mod("torchvision.models.mobilenetv3", mobilenet_v3_large=lambda *a, **k: None)
    _mod("torchvision.models.resnet", resnet50=lambda *a, **k: None)
    _mod("torchvision.models.detection")
    _mod("torchvision.models.detection._utils", BoxCoder=TV.BoxCoder, Matcher=TV.Matcher,
         BalancedPositiveNegativeSampler=TV.BalancedPositiveNegativeSampler, overwrite_eps=lambda *a, **k: None)
    _mod("torchvision.models.detection.anchor_utils", AnchorGenerator=TV.AnchorGenerator)
    _mod("torchvision.models.detection.image_list", ImageList=TV.ImageList)
    _mod("torchvision.models.detection.backbone_utils", _resnet_fpn_extractor=None,
         _validate_trainable_layers=None, _mobilenet_extractor=None, resnet_fpn_backbone=None)
    _mod("torchvision.models.detection.transform", GeneralizedRCNNTransform=_Inert)
    _mod("torchvision._internally_replaced_utils", load_state_dict_from_url=None)
    _mod("torchvision.utils", _log_api_usage_once=lambda *a, **k: None)
    sys.modules["torchvision.ops"].boxes = sys.modules["torchvision.ops.boxes"]
    sys.modules["torchvision.ops"].misc = sys.modules["torchvision.ops.misc"]


def load_reference():
    install_shims()
    sys.path.insert(0, REF)
    with contextlib.redirect_stdout(io.StringIO()):
        import rpn as ref_rpn                      # noqa
        import faster_rcnn as ref_frcnn            # noqa
    return ref_rpn, ref_frcnn


def load_reference_roi_heads():
    with contextlib.redirect_stdout(io.StringIO()):
        import roi_heads as ref_roi_heads          # noqa
    return ref_roi_heads


class _StoredHead(torch.nn.Module):
    """stands in for RPNHeadSNN inside the reference's RegionProposalNetwork: returns stored head outputs (rpn.py:613)"""
    def __init__(self, objectness, deltas):
        super().__init__()
        self.objectness, self.deltas = objectness, deltas

    def forward(self, features):
        return self.objectness, self.deltas


def _save_lists(d, key, tensors):
    d[key + "_n"] = np.array([int(t.shape[0]) for t in tensors], dtype=np.int64)
    width = max([int(np.prod(t.shape[1:])) for t in tensors] + [1])
    d[key] = np_(torch.cat([t.reshape(t.shape[0], width) for t in tensors], 0)) if tensors else np.zeros((0, width))


def gen_rpn_post(ref_rpn, name, spec, out):
    """runs the reference's own RegionProposalNetwork.forward (rpn.py:563-703: anchors, concat_box_prediction_layers,
    BoxCoder.decode, filter_proposals) in eval mode on stored head outputs"""
    obj, dl = FX.rpn_post_inputs(spec)
    N = len(spec["image_sizes"])
    rpn = ref_rpn.RegionProposalNetwork(TV.AnchorGenerator(FX.ANCHOR_SIZES, FX.ASPECT_RATIOS), _StoredHead(obj, dl), 0.7, 0.3, 256, 0.5,
                                        dict(training=2000, testing=spec["pre"]), dict(training=2000, testing=spec["post"]),
                                        spec["nms"], score_thresh=spec["score_thresh"]).eval()
    images = TV.ImageList(torch.zeros((N, 3) + tuple(spec["canvas"])), list(spec["image_sizes"]))
    feats = {str(l): torch.zeros((N, 1) + tuple(g)) for l, g in enumerate(spec["grids"])}
    with torch.no_grad():
        boxes, extras = rpn(images, feats)                    # the reference's own forward
        scores = rpn.filter_proposals(                        # (forward drops the scores; same body called once more for them)
            TV.BoxCoder((1.0, 1.0, 1.0, 1.0)).decode(ref_rpn.concat_box_prediction_layers(obj, dl)[1], rpn.anchor_generator(images, list(feats.values()))).view(N, -1, 4),
            ref_rpn.concat_box_prediction_layers(obj, dl)[0], images.image_sizes, [o.shape[1] * o.shape[2] * o.shape[3] for o in obj])[1]
    st = {}
    o_b, o_s, o_pre = PO.rpn_proposals(obj, dl, spec["canvas"], spec["image_sizes"], FX.ANCHOR_SIZES, FX.ASPECT_RATIOS, spec["pre"],
                                       spec["post"], spec["nms"], spec["score_thresh"], stats=st)
    for i in range(N):
        assert torch.equal(boxes[i], o_b[i]) and torch.equal(scores[i], o_s[i]), "oracle != reference filter_proposals (%s)" % name
        assert torch.equal(extras[i]["proposals"], o_pre[i]["proposals"]) and torch.equal(extras[i]["objectness"], o_pre[i]["objectness"])
    d = {"min_gap": np.array(st.get("min_gap", 1.0))}
    _save_lists(d, "boxes", boxes)
    _save_lists(d, "scores", scores)
    d["pre_boxes"] = np_(torch.stack([e["proposals"] for e in extras]))
    d["pre_prob"] = np_(torch.stack([e["objectness"] for e in extras]))
    np.savez_compressed(os.path.join(out, name + ".npz"), **d)
    print("wrote %-20s proposals %s of %d candidates, min |IoU - thr| %.2e" % (name, [int(b.shape[0]) for b in boxes],
          extras[0]["proposals"].shape[0], st.get("min_gap", 1.0)))


def gen_det_post(ref_roi, name, spec, out):
    """runs the reference's own RoIHeadsSNN.postprocess_detections (roi_heads.py:1075-1176)"""
    logits, reg, props = FX.det_post_inputs(spec)
    heads = ref_roi.RoIHeadsSNN(None, None, 0.5, 0.5, 512, 0.25, None, 0.4, 0.5, 100).eval()      # model.py:98-106
    with torch.no_grad():
        res = heads.postprocess_detections(logits, reg, props, list(spec["image_shapes"]))
    st = {}
    ora = PO.det_postprocess(logits, reg, props, list(spec["image_shapes"]), stats=st)
    for a, b in zip(res, ora):
        for x, y in zip(a, b):
            assert x.dtype == y.dtype and torch.equal(x, y), "oracle != reference postprocess_detections (%s)" % name
    d = {"min_gap": np.array(st.get("min_gap", 1.0))}
    # how many detections the result would have with IoUs on the raw coordinates (torchvision's > 4000-coordinate strategy)
    real = TV.batched_nms
    TV.batched_nms = lambda b, s_, i, t, stats=None: TV._batched_nms_vanilla(b, s_, i, t, stats)
    try:
        raw = PO.det_postprocess(logits, reg, props, list(spec["image_shapes"]))
    finally:
        TV.batched_nms = real
    d["n_raw"] = np.array([int(b.shape[0]) for b in raw[0]], dtype=np.int64)
    for key, lst in zip(("boxes", "scores", "labels", "all_scores", "all_boxes"), res):
        _save_lists(d, key, list(lst))
    np.savez_compressed(os.path.join(out, name + ".npz"), **d)
    print("wrote %-20s detections %s (fg %s), min |IoU - thr| %.2e, with raw-coordinate IoUs %s" % (name, [int(b.shape[0]) for b in res[0]],
          [int((l > 0).sum()) for l in res[2]], st.get("min_gap", 1.0), d["n_raw"].tolist()))


def _exec_literal_forward(path, first, last):
    """exec the spike-rate ``forward`` kept as a string literal in the reference (text read from
    /root/reference at run time, never stored)."""
    with open(path) as f:
        lines = f.readlines()[first - 1:last]
    src = textwrap.dedent("".join(lines))
    from typing import List, Tuple
    from torch import Tensor
    ns = {"torch": torch, "List": List, "Tuple": Tuple, "Tensor": Tensor,
          "lif_current_encoder": NR.lif_current_encoder}
    exec(src, ns)
    return ns["forward"]


def np_(t):
    return t.detach().cpu().numpy()


def gen_rpn(ref_rpn, name, spec, out):
    C, A, T = spec["C"], spec["A"], spec["T"]
    feats, w_shared, w_cls, w_bbox = FX.rpn_inputs(spec)
    with contextlib.redirect_stdout(io.StringIO()):
        head = ref_rpn.RPNHeadSNN(C, A, T)
    sd = head.state_dict()
    assert sorted(sd.keys()) == ["conv_bbox.weight", "conv_cls.weight", "shared_conv.weight"], sd.keys()
    with torch.no_grad():
        head.shared_conv.weight.copy_(w_shared)
        head.conv_cls.weight.copy_(w_cls)
        head.conv_bbox.weight.copy_(w_bbox)
        logits, bbox = head(feats)                       # the reference's own forward, rpn.py:84-121
        rate_fwd = _exec_literal_forward(os.path.join(REF, "rpn.py"), 126, 200)
        l2, b2, rates = rate_fwd(head, feats)            # rpn.py:126-200 verbatim
    # cross-check the oracle restatement bit-for-bit against the reference's own loop
    o_l, o_b, o_r, o_tr = OR.rpn_head_forward(feats, w_shared, w_cls, w_bbox, T, trace=True,
                                               spike_rates=True)
    for a, b in zip(logits + bbox + list(rates), list(o_l) + list(o_b) + list(o_r)):
        assert a.dtype == b.dtype and torch.equal(a, b), "oracle != reference loop (%s)" % name
    for a, b in zip(l2 + b2, logits + bbox):
        assert torch.equal(a, b)
    d = {}
    enc_rate, lif_rate = [], []
    for l in range(len(feats)):
        d["logits%d" % l] = np_(logits[l])
        d["bbox%d" % l] = np_(bbox[l])
        for j in range(3):
            d["rate%d_%d" % (l, j)] = np_(rates[3 * l + j])
        # per-step shared-LIF spikes (bit-packed) from the oracle trace == reference loop values
        d["spk%d" % l] = np.packbits(np_(o_tr[l]["spk"]).astype(np.uint8), axis=None)
        d["spk%d_shape" % l] = np.array(o_tr[l]["spk"].shape, dtype=np.int64)
        enc_rate.append(float(o_tr[l]["z"].mean())); lif_rate.append(float(o_tr[l]["spk"].mean()))
    np.savez_compressed(os.path.join(out, name + ".npz"), **d)
    print("wrote %-20s enc rate %s  shared-LIF rate %s" % (name, np.round(enc_rate, 3), np.round(lif_rate, 3)))


def gen_det(ref_frcnn, name, spec, out):
    C, Hd, K, T = spec["C"], spec["Hd"], spec["K"], spec["T"]
    oob = spec.get("only_one_bbox", False)
    x, w6, w7, w_cls, w_bbox = FX.det_inputs(spec)
    with contextlib.redirect_stdout(io.StringIO()):
        head = ref_frcnn.FastRCNNPredictorSNNFull(C * 49, Hd, K, T, only_one_bbox=oob)
    sd = head.state_dict()
    assert sorted(sd.keys()) == ["bbox_pred.weight", "cls_score.weight", "fc6.weight", "fc7.weight"]
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        head.fc6.weight.copy_(w6); head.fc7.weight.copy_(w7)
        head.cls_score.weight.copy_(w_cls); head.bbox_pred.weight.copy_(w_bbox)
        cls, bbox = head(x)                              # faster_rcnn.py:470-516
        rate_fwd = _exec_literal_forward(os.path.join(REF, "faster_rcnn.py"), 520, 618)
        rates = rate_fwd(head, x)                        # faster_rcnn.py:520-618 verbatim
    o_c, o_b, o_tr = OR.det_head_forward(x, w6, w7, w_cls, w_bbox, T, trace=True)
    o_r = OR.det_head_forward(x, w6, w7, w_cls, w_bbox, T, spike_rates=True, only_one_bbox=oob)
    for a, b in zip([cls, bbox] + list(rates), [o_c, o_b] + list(o_r)):
        assert a.dtype == b.dtype and torch.equal(a, b), "oracle != reference loop (%s)" % name
    d = {"cls": np_(cls), "bbox": np_(bbox),
         "spk6": np.packbits(np_(o_tr["spk6"]).astype(np.uint8), axis=None),
         "spk6_shape": np.array(o_tr["spk6"].shape, dtype=np.int64),
         "spk7": np.packbits(np_(o_tr["spk7"]).astype(np.uint8), axis=None),
         "spk7_shape": np.array(o_tr["spk7"].shape, dtype=np.int64)}
    for j, r in enumerate(rates):
        d["rate%d" % j] = np_(r)
    np.savez_compressed(os.path.join(out, name + ".npz"), **d)
    print("wrote %-20s enc %.3f spk6 %.3f spk7 %.3f" % (name, float(o_tr["z"].mean()),
          float(o_tr["spk6"].mean()), float(o_tr["spk7"].mean())))


def gen_roi(name, spec, out):
    """box pooling: torchvision is absent, so this fixture is written from the RESTATEMENT (oracle/roi_align_oracle.py, pinned by
    closed-form known answers), then the detector-head oracle on the pooled features"""
    from oracle import roi_align_oracle as RA
    feats, boxes, shapes = FX.roi_inputs(spec)
    pooled = RA.multiscale_roi_align(feats, boxes, shapes)
    w6, w7, wc, wb = FX.roi_head_weights(spec)
    cls, bbox = OR.det_head_forward(pooled, w6, w7, wc, wb, spec["T"])
    np.savez_compressed(os.path.join(out, name + ".npz"), pooled=np_(pooled), cls=np_(cls), bbox=np_(bbox))
    print("wrote %-20s pooled %s |mean| %.3f" % (name, tuple(pooled.shape), float(pooled.abs().mean())))


def gen_energy(name, spec, out):
    """execs the reference's own energy block (train.py:472-515: text read from /root/reference at run time, never stored) on the
    fixture's rate lists; stores per layer (mean spikes over T, FLOPs) and the two totals"""
    import types as _types
    rates = FX.energy_rates(spec)
    with open(os.path.join(REF, "train.py")) as f:
        src = textwrap.dedent("".join(f.readlines()[471:515]))
    model = _types.SimpleNamespace(rpn=_types.SimpleNamespace(head=_types.SimpleNamespace(num_steps=spec["T_rpn"])),
                                   roi_heads=_types.SimpleNamespace(box_head_and_predictor=_types.SimpleNamespace(num_steps=spec["T_det"])))
    ns = {"torch": torch, "model": model, "all_images_per_layer_dict": dict(rates)}
    with contextlib.redirect_stdout(io.StringIO()):
        exec(src, ns)
    per_layer = np.array([[float(a), float(b)] for a, b in ns["flops_per_layer"]], dtype=np.float64)
    d = {"per_layer": per_layer, "ann_total": np.array(float(ns["ann_total_energy_consumption"])),
         "snn_total": np.array(float(ns["snn_total_energy_consumption"])),
         "layer_names": np.array(ns["all_layers_names"][:len(per_layer)])}
    np.savez_compressed(os.path.join(out, name + ".npz"), **d)
    print("wrote %-20s layers %d  SNN / ANN energy %.4f" % (name, len(per_layer), float(d["snn_total"]) / float(d["ann_total"])))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=FX.GOLDEN_DIR)
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    torch.set_num_threads(8)
    ref_rpn, ref_frcnn = load_reference()
    for name, spec in FX.RPN_SPECS.items():
        gen_rpn(ref_rpn, name, spec, args.out)
    for name, spec in FX.DET_SPECS.items():
        gen_det(ref_frcnn, name, spec, args.out)
    ref_roi = load_reference_roi_heads()
    for name, spec in FX.RPN_POST_SPECS.items():
        gen_rpn_post(ref_rpn, name, spec, args.out)
    for name, spec in FX.DET_POST_SPECS.items():
        gen_det_post(ref_roi, name, spec, args.out)
    for name, spec in FX.ROI_SPECS.items():
        gen_roi(name, spec, args.out)
    for name, spec in FX.ENERGY_SPECS.items():
        gen_energy(name, spec, args.out)


if __name__ == "__main__":
    main()
