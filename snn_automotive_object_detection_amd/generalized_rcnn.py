"""``GeneralizedRCNN`` with the reference's signature (/root/reference/generalized_rcnn.py:15-170,
inference side): transform -> backbone (no_grad) -> rpn -> roi_heads -> postprocess; in eval the per-image
``proposals`` / ``objectness`` of the RPN are merged into the detections (125-132) and ``all_boxes`` /
``proposals`` are mapped back to the original image size (146-170)."""
from collections import OrderedDict
from typing import Dict, List, Tuple

import torch
from torch import nn, Tensor

from .stock.transform import resize_boxes


class GeneralizedRCNN(nn.Module):
    def __init__(self, backbone: nn.Module, rpn: nn.Module, roi_heads: nn.Module, transform: nn.Module) -> None:
        super().__init__()
        self.transform = transform
        self.backbone = backbone
        self.rpn = rpn
        self.roi_heads = roi_heads

    @torch.no_grad()
    def forward(self, images: List[Tensor], targets=None):
        if self.training:
            raise NotImplementedError("inference only (call .eval()): training is out of scope (DESIGN.md §7)")
        original_image_sizes = [(int(img.shape[-2]), int(img.shape[-1])) for img in images]
        images, targets = self.transform(images, targets)
        features = self.backbone(images.tensors)                                    # generalized_rcnn.py:93-94
        if isinstance(features, torch.Tensor):
            features = OrderedDict([("0", features)])
        head = getattr(self.rpn, "head", None)
        if getattr(head, "spike_rates", False):                                     # spike-rate path, 98-111: ONE head call;
            proposals, rpn_rates = self.rpn(images, features, targets)              # the RPN hands the head's rates on (rpn.py:698-701)
            det_rates = self.roi_heads(features, proposals, images.image_sizes, targets)
            return list(rpn_rates) + list(det_rates)
        proposals, proposal_extras = self.rpn(images, features, targets)            # :114
        detections, _ = self.roi_heads(features, proposals, images.image_sizes, targets)   # :118
        detections = self.transform.postprocess(detections, images.image_sizes, original_image_sizes)
        for i in range(len(detections)):                                            # :125-129
            for k, v in proposal_extras[i].items():
                detections[i][k] = v
        if detections and "all_boxes" in detections[0]:
            detections = self.postprocess(detections, images.image_sizes, original_image_sizes)
        return detections

    def postprocess(self, result: List[Dict[str, Tensor]], image_shapes: List[Tuple[int, int]],
                    original_image_sizes: List[Tuple[int, int]]) -> List[Dict[str, Tensor]]:
        for i, (pred, im_s, o_im_s) in enumerate(zip(result, image_shapes, original_image_sizes)):
            boxes = pred["all_boxes"]
            shape = boxes.shape
            result[i]["all_boxes"] = resize_boxes(boxes.reshape(shape[0] * shape[1], -1), im_s, o_im_s).reshape(*shape)
            if "proposals" in pred:
                result[i]["proposals"] = resize_boxes(pred["proposals"], im_s, o_im_s)
        return result
