"""ORACLE — test infrastructure only (imported by tests/, oracle/make_golden.py; never by the product package).

CPU restatement of the torchvision 0.13.1 pieces the reference's RPN / RoI-head post-processing calls
(README.md:17 pins the version; the package is absent here and from /root/reference, so this is "third-party
algorithm restated" - pinned by the reference's own call sites):

  box_ops.clip_boxes_to_image / remove_small_boxes / batched_nms / nms     rpn.py:505-517, roi_heads.py:1100,1151-1161
  det_utils.BoxCoder(weights).decode                                       rpn.py:347,663; roi_heads.py:938-940,1087
  AnchorGenerator(sizes, aspect_ratios)(image_list, feature_maps)          faster_rcnn.py:31-34, rpn.py:636
  ImageList                                                                rpn.py:14

``nms`` follows torchvision's CPU kernel (csrc/ops/cpu/nms_kernel.cpp): boxes visited by decreasing score, a kept
box suppresses every later one whose IoU with it is > threshold, IoU = inter / (area_i + area_j - inter) in fp32 with
that operation order.  ``batched_nms`` keeps torchvision's two strategies and its CPU switch-over: more than 4000
box coordinates -> one NMS per category ("vanilla"); otherwise the coordinate trick (every category shifted into its own
coordinate range, one NMS).  The trick rounds the shifted coordinates, so an IoU that sits within ~1e-6 of the threshold can
go either way between the two strategies; the fixtures record the smallest |IoU - threshold| they contain.
Independent of the product's stock/ glue and of the HIP NMS on purpose."""
import math
from typing import List, Tuple

import numpy as np
import torch
from torch import Tensor


class ImageList:
    def __init__(self, tensors: Tensor, image_sizes: List[Tuple[int, int]]):
        self.tensors = tensors
        self.image_sizes = image_sizes


def _is_tracing() -> bool:
    return False


# ---------------------------------------------------------------------------------------------
# torchvision.ops.boxes
# ---------------------------------------------------------------------------------------------
def clip_boxes_to_image(boxes: Tensor, size: Tuple[int, int]) -> Tensor:
    height, width = size
    xs = boxes[..., 0::2].clamp(min=0, max=width)
    ys = boxes[..., 1::2].clamp(min=0, max=height)
    return torch.stack((xs, ys), dim=boxes.dim()).reshape(boxes.shape)


def remove_small_boxes(boxes: Tensor, min_size: float) -> Tensor:
    w = boxes[:, 2] - boxes[:, 0]
    h = boxes[:, 3] - boxes[:, 1]
    return torch.where((w >= min_size) & (h >= min_size))[0]


def box_iou(a: Tensor, b: Tensor) -> Tensor:
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = torch.max(a[:, None, :2], b[None, :, :2])
    rb = torch.min(a[:, None, 2:], b[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    return inter / (area_a[:, None] + area_b[None, :] - inter)


def nms(boxes: Tensor, scores: Tensor, iou_threshold: float, stats: dict = None) -> Tensor:
    """greedy NMS of torchvision's CPU kernel; returns kept indices by decreasing score (int64).
    Equal scores are visited in index order (stable sort; the C++ kernel's sort leaves that order unspecified).
    ``stats``: optional dict; 'min_gap' receives the smallest |IoU - threshold| over the pairs that were compared."""
    n = int(boxes.shape[0])
    if n == 0:
        return torch.empty((0,), dtype=torch.int64)
    b = boxes.detach().to(torch.float32).cpu().numpy()
    s = scores.detach().to(torch.float32).cpu().numpy()
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    areas = (x2 - x1) * (y2 - y1)                                  # fp32, like the kernel's areas_t
    order = np.argsort(-s, kind="stable")
    suppressed = np.zeros(n, dtype=bool)
    keep = []
    thr = np.float64(iou_threshold)                                # the kernel compares a float IoU with a double threshold
    zero = np.float32(0)
    for _i in range(n):
        i = order[_i]
        if suppressed[i]:
            continue
        keep.append(i)
        rest = order[_i + 1:]
        rest = rest[~suppressed[rest]]
        if rest.size == 0:
            continue
        w = np.maximum(zero, np.minimum(x2[i], x2[rest]) - np.maximum(x1[i], x1[rest]))
        h = np.maximum(zero, np.minimum(y2[i], y2[rest]) - np.maximum(y1[i], y1[rest]))
        inter = w * h
        with np.errstate(invalid="ignore", divide="ignore"):
            ovr = inter / ((areas[i] + areas[rest]) - inter)
        if stats is not None:
            gap = np.abs(ovr.astype(np.float64) - thr)
            gap = gap[np.isfinite(gap)]
            if gap.size:
                stats["min_gap"] = min(stats.get("min_gap", 1.0), float(gap.min()))
        suppressed[rest[ovr.astype(np.float64) > thr]] = True
    return torch.from_numpy(np.asarray(keep, dtype=np.int64))


def _batched_nms_coordinate_trick(boxes, scores, idxs, iou_threshold, stats=None):
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)
    max_coordinate = boxes.max()
    offsets = idxs.to(boxes) * (max_coordinate + torch.tensor(1).to(boxes))
    return nms(boxes + offsets[:, None], scores, iou_threshold, stats)


def _batched_nms_vanilla(boxes, scores, idxs, iou_threshold, stats=None):
    keep_mask = torch.zeros_like(scores, dtype=torch.bool)
    for class_id in torch.unique(idxs):
        cur = torch.where(idxs == class_id)[0]
        keep_mask[cur[nms(boxes[cur], scores[cur], iou_threshold, stats)]] = True
    keep = torch.where(keep_mask)[0]
    return keep[scores[keep].sort(descending=True, stable=True)[1]]


def batched_nms(boxes: Tensor, scores: Tensor, idxs: Tensor, iou_threshold: float, stats: dict = None) -> Tensor:
    if boxes.numel() > 4000:                                      # torchvision's CPU threshold
        return _batched_nms_vanilla(boxes, scores, idxs, iou_threshold, stats)
    return _batched_nms_coordinate_trick(boxes, scores, idxs, iou_threshold, stats)


# ---------------------------------------------------------------------------------------------
# torchvision.models.detection._utils
# ---------------------------------------------------------------------------------------------
class BoxCoder:
    def __init__(self, weights, bbox_xform_clip: float = math.log(1000.0 / 16)):
        self.weights = weights
        self.bbox_xform_clip = bbox_xform_clip

    def decode(self, rel_codes: Tensor, boxes: List[Tensor]) -> Tensor:
        per_image = [int(b.size(0)) for b in boxes]
        concat = torch.cat(list(boxes), dim=0)
        total = sum(per_image)
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
        dw = rel_codes[:, 2::4] / ww
        dh = rel_codes[:, 3::4] / wh
        dw = torch.clamp(dw, max=self.bbox_xform_clip)            # exp() of a huge delta would overflow
        dh = torch.clamp(dh, max=self.bbox_xform_clip)
        pcx = dx * widths[:, None] + ctr_x[:, None]
        pcy = dy * heights[:, None] + ctr_y[:, None]
        pw = torch.exp(dw) * widths[:, None]
        ph = torch.exp(dh) * heights[:, None]
        hh = torch.tensor(0.5, dtype=pcy.dtype) * ph
        hw = torch.tensor(0.5, dtype=pcx.dtype) * pw
        return torch.stack((pcx - hw, pcy - hh, pcx + hw, pcy + hh), dim=2).flatten(1)


class Matcher:                                                    # training only; constructed by the reference's __init__
    BELOW_LOW_THRESHOLD = -1
    BETWEEN_THRESHOLDS = -2

    def __init__(self, high_threshold, low_threshold, allow_low_quality_matches=False):
        self.high_threshold, self.low_threshold = high_threshold, low_threshold
        self.allow_low_quality_matches = allow_low_quality_matches


class BalancedPositiveNegativeSampler:                            # training only
    def __init__(self, batch_size_per_image, positive_fraction):
        self.batch_size_per_image, self.positive_fraction = batch_size_per_image, positive_fraction


# ---------------------------------------------------------------------------------------------
# torchvision.models.detection.anchor_utils
# ---------------------------------------------------------------------------------------------
class AnchorGenerator(torch.nn.Module):
    def __init__(self, sizes=((128, 256, 512),), aspect_ratios=((0.5, 1.0, 2.0),)):
        super().__init__()
        if not isinstance(sizes[0], (list, tuple)):
            sizes = tuple((s,) for s in sizes)
        if not isinstance(aspect_ratios[0], (list, tuple)):
            aspect_ratios = (aspect_ratios,) * len(sizes)
        self.sizes, self.aspect_ratios = sizes, aspect_ratios
        self.cell_anchors = [self.generate_anchors(s, a) for s, a in zip(sizes, aspect_ratios)]

    @staticmethod
    def generate_anchors(scales, aspect_ratios, dtype=torch.float32):
        scales = torch.as_tensor(scales, dtype=dtype)
        aspect_ratios = torch.as_tensor(aspect_ratios, dtype=dtype)
        h_ratios = torch.sqrt(aspect_ratios)
        w_ratios = 1 / h_ratios
        ws = (w_ratios[:, None] * scales[None, :]).view(-1)
        hs = (h_ratios[:, None] * scales[None, :]).view(-1)
        return (torch.stack([-ws, -hs, ws, hs], dim=1) / 2).round()

    def num_anchors_per_location(self):
        return [len(s) * len(a) for s, a in zip(self.sizes, self.aspect_ratios)]

    def forward(self, image_list: ImageList, feature_maps: List[Tensor]) -> List[Tensor]:
        grid_sizes = [fm.shape[-2:] for fm in feature_maps]
        image_size = image_list.tensors.shape[-2:]
        dtype = feature_maps[0].dtype
        per_level = []
        for (gh, gw), base in zip(grid_sizes, self.cell_anchors):
            sh, sw = image_size[0] // gh, image_size[1] // gw
            shifts_x = torch.arange(0, gw, dtype=torch.int32) * sw
            shifts_y = torch.arange(0, gh, dtype=torch.int32) * sh
            sy, sx = torch.meshgrid(shifts_y, shifts_x, indexing="ij")
            sx, sy = sx.reshape(-1), sy.reshape(-1)
            shifts = torch.stack((sx, sy, sx, sy), dim=1)
            per_level.append((shifts.view(-1, 1, 4) + base.to(dtype).view(1, -1, 4)).reshape(-1, 4))
        return [torch.cat(per_level) for _ in image_list.image_sizes]
