"""``RoIHeadsSNN`` with the reference's constructor/forward signature (/root/reference/roi_heads.py:
901-1347, inference side): box_roi_pool -> **spiking head** (the accelerated call, roi_heads.py:1230) ->
postprocess_detections (1075-1176) that keeps the surviving background boxes and returns
``all_scores`` / ``all_boxes`` for new-object discovery.  Everything except the head call is stock torch."""
import warnings
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import nn, Tensor

from .stock import boxes as box_ops
from .rpn import _WARN_LOCK


class RoIHeadsSNN(nn.Module):
    def __init__(self, box_roi_pool, box_head_and_predictor,
                 fg_iou_thresh, bg_iou_thresh, batch_size_per_image, positive_fraction, bbox_reg_weights,
                 score_thresh, nms_thresh, detections_per_img,
                 mask_roi_pool=None, mask_head=None, mask_predictor=None,
                 keypoint_roi_pool=None, keypoint_head=None, keypoint_predictor=None):
        super().__init__()
        if bbox_reg_weights is None:
            bbox_reg_weights = (10.0, 10.0, 5.0, 5.0)                                # roi_heads.py:938-940
        self.box_coder = box_ops.BoxCoder(bbox_reg_weights)
        self.box_roi_pool = box_roi_pool
        self.box_head_and_predictor = box_head_and_predictor
        self.score_thresh = score_thresh
        self.nms_thresh = nms_thresh
        self.detections_per_img = detections_per_img
        self.fuse_roi_align = True        # use the fused RoIAlign+encoder kernel when pool/head support it
        self.post = "hip"                 # "hip": snn_det_postprocess; "reference": the reference's order on stock torch ops
        self._warned = set()              # fallback reasons already reported
        # training-side hyper-parameters are accepted for signature compatibility only
        self.fg_iou_thresh, self.bg_iou_thresh = fg_iou_thresh, bg_iou_thresh
        self.batch_size_per_image, self.positive_fraction = batch_size_per_image, positive_fraction
        if any(m is not None for m in (mask_roi_pool, mask_head, mask_predictor, keypoint_roi_pool,
                                       keypoint_head, keypoint_predictor)):
            raise NotImplementedError("mask/keypoint branches are dead code in the reference (train.py:263-268)")

    def has_mask(self):
        return False

    def has_keypoint(self):
        return False

    def postprocess_detections(self, class_logits: Tensor, box_regression: Tensor, proposals: List[Tensor],
                               image_shapes: List[Tuple[int, int]]):
        per_image = [p.shape[0] for p in proposals]
        if box_regression.shape[-1] != 4 * class_logits.shape[-1]:
            # --only-one-bbox heads (faster_rcnn.py:460-467) give [R, 4]: the reference's own post-processing then slices
            # boxes[:, 1:] (roi_heads.py:1115) down to nothing while scores keep K-1 columns - not a usable path
            raise NotImplementedError("postprocess_detections needs %d box outputs per RoI (4 per class), got %d "
                                      "(only_one_bbox heads are not supported past the head, SURVEY.md appendix C.6)"
                                      % (4 * class_logits.shape[-1], box_regression.shape[-1]))
        if class_logits.is_cuda and self.post == "hip" and per_image and max(per_image) > 0:
            why = self._hip_postprocess_refusal(len(per_image), max(per_image), class_logits.shape[-1])
            if why is None:
                return self._postprocess_hip(class_logits, box_regression, proposals, image_shapes, per_image)
            with _WARN_LOCK:                                  # loud, once per reason: the stock-torch path is ~4x slower per batch
                first = why not in self._warned
                self._warned.add(why)
            if first:
                warnings.warn("RoIHeadsSNN: detection post-processing falls back to the stock torch ops (%s)" % why, RuntimeWarning)
        return self.postprocess_detections_reference(class_logits, box_regression, proposals, image_shapes)

    def _hip_postprocess_refusal(self, n_images: int, max_rois: int, n_classes: int):
        """the limits of snn_det_postprocess (csrc/snn_kernels.hip / snn_post.h), mirrored so that a configuration outside them
        takes the reference path instead of raising: None if the HIP path can run, else the reason"""
        if n_images > 64:
            return "%d images per batch (HIP path: <= 64)" % n_images
        if n_classes < 2 or n_classes > 96:
            return "%d classes (HIP path: 2..96)" % n_classes
        if max_rois > 10240:                                  # 2 x 64 x ceil(R / 64) x 8 B of LDS for the NMS walk (also <= DET_SORT_MAX)
            return "%d RoIs per image (HIP path: <= 10240)" % max_rois
        if (n_classes - 1) * min(int(self.detections_per_img), max_rois) > 8192:
            return "(K-1) x detections_per_img = %d ranked candidates per image (HIP path: <= 8192)" % (
                (n_classes - 1) * min(int(self.detections_per_img), max_rois))
        return None

    def _postprocess_hip(self, class_logits, box_regression, proposals, image_shapes, per_image):
        """snn_det_postprocess: five launches and one host synchronisation for the batch"""
        from . import ops
        boxes, scores, labels, counts, all_scores, all_boxes = ops.det_postprocess(
            class_logits.detach(), box_regression.detach(), torch.cat(proposals, 0), per_image, image_shapes,
            self.box_coder.weights, self.score_thresh, self.nms_thresh, self.detections_per_img)
        cnt = counts.sum(1).tolist()                                                   # the one host synchronisation
        out = ([], [], [], list(all_scores.split(per_image, 0)), list(all_boxes.split(per_image, 0)))
        for i, c in enumerate(cnt):
            out[0].append(boxes[i, :c]); out[1].append(scores[i, :c]); out[2].append(labels[i, :c].to(torch.int64))
        return out

    def postprocess_detections_reference(self, class_logits: Tensor, box_regression: Tensor, proposals: List[Tensor],
                                         image_shapes: List[Tuple[int, int]]):
        device = class_logits.device
        num_classes = class_logits.shape[-1]
        per_image = [p.shape[0] for p in proposals]
        pred_boxes = self.box_coder.decode(box_regression, proposals)
        pred_scores = F.softmax(class_logits, -1)
        out = ([], [], [], [], [])
        for boxes, scores, shape in zip(pred_boxes.split(per_image, 0), pred_scores.split(per_image, 0), image_shapes):
            boxes = box_ops.clip_boxes_to_image(boxes, shape)
            labels = torch.arange(num_classes, device=device).view(1, -1).expand_as(scores)
            boxes_all, scores_all = boxes.detach().clone(), scores.detach().clone()
            boxes_bg, scores_bg, labels_bg = boxes[:, 0].reshape(-1, 4), scores[:, 0].reshape(-1), labels[:, 0].reshape(-1)
            boxes, scores, labels = boxes[:, 1:].reshape(-1, 4), scores[:, 1:].reshape(-1), labels[:, 1:].reshape(-1)
            inds = torch.where(scores > self.score_thresh)[0]                         # roi_heads.py:1134
            boxes, scores, labels = boxes[inds], scores[inds], labels[inds]
            # background boxes survive only for RoIs without any foreground detection (1137-1148;
            # vectorised form of the reference's per-detection Python loop)
            is_bg = torch.ones(scores_bg.shape[0], dtype=torch.bool, device=device)
            is_bg[torch.div(inds, num_classes - 1, rounding_mode="trunc")] = False
            inds_bg = torch.where(is_bg)[0]
            boxes_bg, scores_bg, labels_bg = boxes_bg[inds_bg], scores_bg[inds_bg], labels_bg[inds_bg]
            keep = box_ops.remove_small_boxes(boxes, min_size=1e-2)
            boxes, scores, labels = boxes[keep], scores[keep], labels[keep]
            keep_bg = box_ops.remove_small_boxes(boxes_bg, min_size=1e-2)
            boxes_bg, scores_bg, labels_bg = boxes_bg[keep_bg], scores_bg[keep_bg], labels_bg[keep_bg]
            keep = box_ops.batched_nms(boxes, scores, labels, self.nms_thresh)[: self.detections_per_img]
            keep_bg = box_ops.batched_nms(boxes_bg, scores_bg, labels_bg, self.nms_thresh)
            out[0].append(torch.cat((boxes[keep], boxes_bg[keep_bg]), dim=0))
            out[1].append(torch.cat((scores[keep], scores_bg[keep_bg]), dim=0))
            out[2].append(torch.cat((labels[keep], labels_bg[keep_bg]), dim=0))
            out[3].append(scores_all)
            out[4].append(boxes_all)
        return out

    def forward(self, features: Dict[str, Tensor], proposals: List[Tensor], image_shapes: List[Tuple[int, int]],
                targets: Optional[List[Dict[str, Tensor]]] = None):
        if self.training:
            raise NotImplementedError("inference only: training the RoI heads is out of scope (DESIGN.md §7)")
        pool, head = self.box_roi_pool, self.box_head_and_predictor
        if (self.fuse_roi_align and hasattr(head, "forward_roialign") and hasattr(pool, "assign")
                and tuple(pool.output_size) == (7, 7) and pool.sampling_ratio == 2):
            # RoIAlign fused into the head's encoder kernel: same values as the two calls below
            feats, scales, rois, lvl = pool.assign(features, proposals, image_shapes)
            head_out = head.forward_roialign(feats, scales, rois, lvl)
        else:
            box_features = pool(features, proposals, image_shapes)                   # roi_heads.py:1217
            head_out = head(box_features)                                            # roi_heads.py:1230 (HIP)
        if getattr(self.box_head_and_predictor, "spike_rates", False):
            return head_out                                                          # roi_heads.py:1219-1223
        class_logits, box_regression = head_out
        boxes, scores, labels, all_scores, all_boxes = self.postprocess_detections(
            class_logits, box_regression, proposals, image_shapes)
        result = [{"boxes": boxes[i], "labels": labels[i], "scores": scores[i], "all_scores": all_scores[i],
                   "all_boxes": all_boxes[i]} for i in range(len(boxes))]
        return result, {}
