"""GeneralizedRCNNTransform(768, 1536, mean, std) as the reference uses it (faster_rcnn.py:163-164,315;
generalized_rcnn.py:80,122): normalise, resize (bilinear, recompute_scale_factor), zero-pad the batch to a
multiple of 32, and map boxes back to the original image size."""
import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import nn, Tensor

from .anchors import ImageList


def resize_boxes(boxes: Tensor, original_size, new_size) -> Tensor:
    rh, rw = [torch.tensor(s, dtype=torch.float32, device=boxes.device) /
              torch.tensor(o, dtype=torch.float32, device=boxes.device) for s, o in zip(new_size, original_size)]
    xmin, ymin, xmax, ymax = boxes.unbind(1)
    return torch.stack((xmin * rw, ymin * rh, xmax * rw, ymax * rh), dim=1)


class GeneralizedRCNNTransform(nn.Module):
    def __init__(self, min_size: int, max_size: int, image_mean, image_std, size_divisible: int = 32):
        super().__init__()
        self.min_size, self.max_size = min_size, max_size
        self.image_mean, self.image_std = image_mean, image_std
        self.size_divisible = size_divisible

    def forward(self, images: List[Tensor], targets: Optional[List[Dict[str, Tensor]]] = None):
        images = [self.resize(self.normalize(img)) for img in images]
        sizes = [tuple(int(s) for s in img.shape[-2:]) for img in images]
        d = self.size_divisible
        hmax = int(math.ceil(max(s[0] for s in sizes) / d) * d)
        wmax = int(math.ceil(max(s[1] for s in sizes) / d) * d)
        batch = images[0].new_zeros((len(images), images[0].shape[0], hmax, wmax))
        for i, img in enumerate(images):
            batch[i, :, : img.shape[1], : img.shape[2]].copy_(img)
        return ImageList(batch, sizes), targets

    def normalize(self, image: Tensor) -> Tensor:
        if not image.is_floating_point():
            raise TypeError("expected a float image tensor, got %s" % image.dtype)
        mean = torch.as_tensor(self.image_mean, dtype=image.dtype, device=image.device)
        std = torch.as_tensor(self.image_std, dtype=image.dtype, device=image.device)
        return (image - mean[:, None, None]) / std[:, None, None]

    def resize(self, image: Tensor) -> Tensor:
        h, w = image.shape[-2:]
        scale = min(self.min_size / min(h, w), self.max_size / max(h, w))
        return F.interpolate(image[None], scale_factor=scale, mode="bilinear", recompute_scale_factor=True,
                             align_corners=False)[0]

    def postprocess(self, result: List[Dict[str, Tensor]], image_shapes, original_image_sizes):
        for i, (pred, im_s, o_im_s) in enumerate(zip(result, image_shapes, original_image_sizes)):
            result[i]["boxes"] = resize_boxes(pred["boxes"], im_s, o_im_s)
        return result
