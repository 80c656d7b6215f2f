"""Box utilities in plain torch (reference call sites: rpn.py:505-517, roi_heads.py:1087,1100,1151-1161)."""
import math
from typing import List, Tuple

import torch
from torch import Tensor


def box_area(b: Tensor) -> Tensor:
    return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])


def box_iou(a: Tensor, b: Tensor) -> Tensor:
    area_a, area_b = box_area(a), box_area(b)
    lt = torch.max(a[:, None, :2], b[None, :, :2])
    rb = torch.min(a[:, None, 2:], b[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    return inter / (area_a[:, None] + area_b[None, :] - inter)


def clip_boxes_to_image(boxes: Tensor, size: Tuple[int, int]) -> Tensor:
    h, w = size
    x = boxes[..., 0::2].clamp(min=0, max=w)
    y = boxes[..., 1::2].clamp(min=0, max=h)
    return torch.stack((x, y), dim=boxes.dim()).reshape(boxes.shape)


def remove_small_boxes(boxes: Tensor, min_size: float) -> Tensor:
    ws, hs = boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]
    return torch.where((ws >= min_size) & (hs >= min_size))[0]


def nms(boxes: Tensor, scores: Tensor, iou_threshold: float) -> Tensor:
    """Greedy NMS; returns kept indices sorted by decreasing score.  Data-parallel formulation: with
    boxes sorted by score, keep[i] = not any_{j<i}(keep[j] and iou[j,i] > thr); iterating that map from
    keep = all-true reaches its unique fixed point (= the greedy result: element k is final after k
    iterations, in practice a handful)."""
    n = boxes.shape[0]
    if n == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    order = scores.argsort(descending=True, stable=True)
    b = boxes[order]
    sup = torch.triu(box_iou(b, b) > iou_threshold, diagonal=1)          # sup[j, i]: j (better) suppresses i
    supf = sup.to(torch.float32)
    keep = torch.ones(n, dtype=torch.bool, device=boxes.device)
    for _ in range(n):
        new_keep = (keep.to(torch.float32) @ supf) == 0
        if torch.equal(new_keep, keep):
            break
        keep = new_keep
    return order[keep]


HIP_NMS = True     # False: GPU tensors also take the plain-torch NMS (tests: a comparator independent of the HIP kernels)


def batched_nms(boxes: Tensor, scores: Tensor, idxs: Tensor, iou_threshold: float) -> Tensor:
    """NMS per category.  GPU tensors go to the HIP kernels (ops.batched_nms: bit-matrix + in-order scan, IoU on the raw
    coordinates).  Otherwise torchvision's two strategies with its CPU switch-over (the reference's CPU path is the parity
    target): more than 4000 coordinates -> one NMS per category on the raw coordinates; else every category is shifted into
    its own coordinate range and one NMS runs over all of them (the shift rounds the coordinates, so an IoU within ~1e-6 of the
    threshold can be decided differently by the two)."""
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    if boxes.is_cuda and boxes.shape[0] <= 16384 and HIP_NMS:
        from .. import ops
        return ops.batched_nms(boxes, scores, idxs, iou_threshold)
    if boxes.numel() > 4000:
        keep_mask = torch.zeros_like(scores, dtype=torch.bool)
        for c in torch.unique(idxs):
            cur = torch.where(idxs == c)[0]
            keep_mask[cur[nms(boxes[cur], scores[cur], iou_threshold)]] = True
        keep = torch.where(keep_mask)[0]
        return keep[scores[keep].sort(descending=True, stable=True)[1]]
    max_coordinate = boxes.max()
    offsets = idxs.to(boxes) * (max_coordinate + torch.tensor(1).to(boxes))
    return nms(boxes + offsets[:, None], scores, iou_threshold)


class BoxCoder:
    """decode side of torchvision's BoxCoder (rpn.py:347,663 weights (1,1,1,1); roi_heads.py:938-940,1087
    weights (10,10,5,5))."""

    def __init__(self, weights: Tuple[float, float, float, float], bbox_xform_clip: float = math.log(1000.0 / 16)):
        self.weights = weights
        self.bbox_xform_clip = bbox_xform_clip

    def decode(self, rel_codes: Tensor, boxes: List[Tensor]) -> Tensor:
        concat = torch.cat(list(boxes), dim=0)
        total = concat.shape[0]
        if total > 0:
            rel_codes = rel_codes.reshape(total, -1)
        pred = self.decode_single(rel_codes, concat)
        if total > 0:
            pred = pred.reshape(total, -1, 4)
        return pred

    def decode_single(self, rel_codes: Tensor, boxes: Tensor) -> Tensor:
        boxes = boxes.to(rel_codes.dtype)
        widths = boxes[:, 2] - boxes[:, 0]
        heights = boxes[:, 3] - boxes[:, 1]
        ctr_x = boxes[:, 0] + 0.5 * widths
        ctr_y = boxes[:, 1] + 0.5 * heights
        wx, wy, ww, wh = self.weights
        dx = rel_codes[:, 0::4] / wx
        dy = rel_codes[:, 1::4] / wy
        dw = torch.clamp(rel_codes[:, 2::4] / ww, max=self.bbox_xform_clip)
        dh = torch.clamp(rel_codes[:, 3::4] / wh, max=self.bbox_xform_clip)
        pred_ctr_x = dx * widths[:, None] + ctr_x[:, None]
        pred_ctr_y = dy * heights[:, None] + ctr_y[:, None]
        pred_w = torch.exp(dw) * widths[:, None]
        pred_h = torch.exp(dh) * heights[:, None]
        half_h = torch.tensor(0.5, dtype=pred_ctr_y.dtype, device=pred_h.device) * pred_h
        half_w = torch.tensor(0.5, dtype=pred_ctr_x.dtype, device=pred_w.device) * pred_w
        out = torch.stack((pred_ctr_x - half_w, pred_ctr_y - half_h, pred_ctr_x + half_w, pred_ctr_y + half_h), dim=2)
        return out.flatten(1)
