"""ORACLE — test infrastructure only (tests/, oracle/make_golden.py, never the product package).

CPU restatement of the two post-processing steps either side of the spiking heads:

  rpn_proposals(...)        /root/reference/rpn.py:563-703 eval path: anchors -> concat_box_prediction_layers (262-296)
                            -> BoxCoder.decode (663) -> filter_proposals (420-499)
  det_postprocess(...)      /root/reference/roi_heads.py:1075-1176 postprocess_detections (incl. the background boxes
                            kept for new-object discovery, 1110-1148)

oracle/make_golden.py runs the reference's OWN bodies of these functions (its RegionProposalNetwork.forward and
RoIHeadsSNN.postprocess_detections, with torchvision replaced by oracle/torchvision_restated.py) and asserts that this
restatement returns bit-identical tensors; the expected outputs are committed under tests/golden/post_*.npz.  The GPU box
has no /root/reference: there the HIP kernels are compared with this restatement at full size and with the fixtures."""
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor

from . import torchvision_restated as TV


def _flatten_level(layer: Tensor, C: int) -> Tensor:
    """[N, A*C, H, W] -> [N, H*W*A, C]   (rpn.py:246-258)"""
    N, _, H, W = layer.shape
    return layer.view(N, -1, C, H, W).permute(0, 3, 4, 1, 2).reshape(N, -1, C)


def rpn_proposals(objectness: List[Tensor], pred_bbox_deltas: List[Tensor], image_tensor_hw: Tuple[int, int],
                  image_sizes: List[Tuple[int, int]], anchor_sizes, aspect_ratios, pre_nms_top_n: int, post_nms_top_n: int,
                  nms_thresh: float, score_thresh: float = 0.0, min_size: float = 1e-3, stats: dict = None):
    """objectness[l] [N, A, H_l, W_l], pred_bbox_deltas[l] [N, 4A, H_l, W_l] (what the head returns, rpn.py:613).
    Returns (boxes per image, scores per image, [{'proposals', 'objectness'} per image]) like rpn.py:692-703."""
    num_images = objectness[0].shape[0]
    feats = [torch.empty((num_images, 1) + tuple(o.shape[-2:])) for o in objectness]
    images = TV.ImageList(torch.empty((num_images, 3) + tuple(image_tensor_hw)), image_sizes)
    anchors = TV.AnchorGenerator(anchor_sizes, aspect_ratios)(images, feats)                     # rpn.py:636
    num_anchors_per_level = [o.shape[1] * o.shape[2] * o.shape[3] for o in objectness]           # rpn.py:645-646
    A = [d.shape[1] // 4 for d in pred_bbox_deltas]
    obj = torch.cat([_flatten_level(o, o.shape[1] // a) for o, a in zip(objectness, A)], dim=1).flatten(0, -2)
    deltas = torch.cat([_flatten_level(d, 4) for d in pred_bbox_deltas], dim=1).reshape(-1, 4)   # rpn.py:294-295
    proposals = TV.BoxCoder((1.0, 1.0, 1.0, 1.0)).decode(deltas.detach(), anchors).view(num_images, -1, 4)   # rpn.py:663-664
    # ---- filter_proposals, rpn.py:420-499 ----
    obj = obj.detach().reshape(num_images, -1)
    levels = torch.cat([torch.full((n,), i, dtype=torch.int64) for i, n in enumerate(num_anchors_per_level)], 0)
    levels = levels.reshape(1, -1).expand_as(obj)
    top, offset = [], 0
    for ob in obj.split(num_anchors_per_level, 1):                                              # rpn.py:403-416
        k = min(pre_nms_top_n, ob.shape[1])
        top.append(ob.topk(k, dim=1)[1] + offset)
        offset += ob.shape[1]
    top = torch.cat(top, dim=1)
    bi = torch.arange(num_images)[:, None]
    obj, levels, proposals = obj[bi, top], levels[bi, top], proposals[bi, top]
    prob = torch.sigmoid(obj)
    pre_nms = [{"proposals": p, "objectness": prob[i]} for i, p in enumerate(proposals)]        # rpn.py:493-499
    final_boxes, final_scores = [], []
    for boxes, scores, lvl, shape in zip(proposals, prob, levels, image_sizes):
        boxes = TV.clip_boxes_to_image(boxes, shape)
        keep = TV.remove_small_boxes(boxes, min_size)
        boxes, scores, lvl = boxes[keep], scores[keep], lvl[keep]
        keep = torch.where(scores >= score_thresh)[0]
        boxes, scores, lvl = boxes[keep], scores[keep], lvl[keep]
        keep = TV.batched_nms(boxes, scores, lvl, nms_thresh, stats)[:post_nms_top_n]
        final_boxes.append(boxes[keep])
        final_scores.append(scores[keep])
    return final_boxes, final_scores, pre_nms


def det_postprocess(class_logits: Tensor, box_regression: Tensor, proposals: List[Tensor],
                    image_shapes: List[Tuple[int, int]], box_weights=(10.0, 10.0, 5.0, 5.0), score_thresh: float = 0.4,
                    nms_thresh: float = 0.5, detections_per_img: int = 100, stats: dict = None):
    """roi_heads.py:1075-1176.  Returns (boxes, scores, labels, all_scores, all_boxes), each a list per image; per image the
    foreground detections (by decreasing score, at most detections_per_img) are followed by ALL surviving background boxes."""
    num_classes = class_logits.shape[-1]
    per_image = [int(p.shape[0]) for p in proposals]
    pred_boxes = TV.BoxCoder(box_weights).decode(box_regression, proposals)
    pred_scores = F.softmax(class_logits, -1)
    out = ([], [], [], [], [])
    for boxes, scores, shape in zip(pred_boxes.split(per_image, 0), pred_scores.split(per_image, 0), image_shapes):
        boxes = TV.clip_boxes_to_image(boxes, shape)
        labels = torch.arange(num_classes).view(1, -1).expand_as(scores)
        boxes_all, scores_all = boxes.detach().clone(), scores.detach().clone()
        boxes_bg, scores_bg, labels_bg = boxes[:, 0].reshape(-1, 4), scores[:, 0].reshape(-1), labels[:, 0].reshape(-1)
        boxes, scores, labels = boxes[:, 1:].reshape(-1, 4), scores[:, 1:].reshape(-1), labels[:, 1:].reshape(-1)
        inds = torch.where(scores > score_thresh)[0]                                            # roi_heads.py:1134
        boxes, scores, labels = boxes[inds], scores[inds], labels[inds]
        # background rows survive only for RoIs none of whose classes passed the threshold (roi_heads.py:1137-1148; the
        # reference walks `inds` in a Python loop and clears a mask entry per detection)
        roi_of = torch.div(inds, num_classes - 1, rounding_mode="trunc")
        mask = torch.ones(scores_bg.shape[0], dtype=torch.bool)
        mask[roi_of] = False
        inds_bg = torch.where(mask)[0]
        boxes_bg, scores_bg, labels_bg = boxes_bg[inds_bg], scores_bg[inds_bg], labels_bg[inds_bg]
        keep = TV.remove_small_boxes(boxes, 1e-2)
        boxes, scores, labels = boxes[keep], scores[keep], labels[keep]
        keep_bg = TV.remove_small_boxes(boxes_bg, 1e-2)
        boxes_bg, scores_bg, labels_bg = boxes_bg[keep_bg], scores_bg[keep_bg], labels_bg[keep_bg]
        keep = TV.batched_nms(boxes, scores, labels, nms_thresh, stats)[:detections_per_img]
        keep_bg = TV.batched_nms(boxes_bg, scores_bg, labels_bg, nms_thresh, stats)
        out[0].append(torch.cat((boxes[keep], boxes_bg[keep_bg]), dim=0))
        out[1].append(torch.cat((scores[keep], scores_bg[keep_bg]), dim=0))
        out[2].append(torch.cat((labels[keep], labels_bg[keep_bg]), dim=0))
        out[3].append(scores_all)
        out[4].append(boxes_all)
    return out
