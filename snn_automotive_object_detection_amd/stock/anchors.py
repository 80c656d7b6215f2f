"""ImageList + AnchorGenerator (faster_rcnn.py:31-34 default: one size x three ratios per level; rpn.py:636)."""
from typing import List, Tuple

import torch
from torch import nn, Tensor

from .._cache import StreamSafeDict


class ImageList:
    def __init__(self, tensors: Tensor, image_sizes: List[Tuple[int, int]]):
        self.tensors = tensors
        self.image_sizes = image_sizes

    def to(self, device):
        return ImageList(self.tensors.to(device), self.image_sizes)


class AnchorGenerator(nn.Module):
    def __init__(self, sizes=((128, 256, 512),), aspect_ratios=((0.5, 1.0, 2.0),)):
        super().__init__()
        if not isinstance(sizes[0], (list, tuple)):
            sizes = tuple((s,) for s in sizes)
        if not isinstance(aspect_ratios[0], (list, tuple)):
            aspect_ratios = (aspect_ratios,) * len(sizes)
        self.sizes = sizes
        self.aspect_ratios = aspect_ratios
        self.cell_anchors = [self._base(s, a) for s, a in zip(sizes, aspect_ratios)]
        self._cache = StreamSafeDict()                  # (grid sizes, image size, dtype, device) -> anchors; safe across threads / streams

    @staticmethod
    def _base(scales, ratios, dtype=torch.float32):
        scales = torch.as_tensor(scales, dtype=dtype)
        ratios = torch.as_tensor(ratios, dtype=dtype)
        h_ratios = torch.sqrt(ratios)
        w_ratios = 1 / h_ratios
        ws = (w_ratios[:, None] * scales[None, :]).view(-1)
        hs = (h_ratios[:, None] * scales[None, :]).view(-1)
        return (torch.stack([-ws, -hs, ws, hs], dim=1) / 2).round()

    def num_anchors_per_location(self) -> List[int]:
        return [len(s) * len(a) for s, a in zip(self.sizes, self.aspect_ratios)]

    def forward(self, image_list: ImageList, feature_maps: List[Tensor]) -> List[Tensor]:
        grid_sizes = [f.shape[-2:] for f in feature_maps]
        image_size = image_list.tensors.shape[-2:]
        dtype, device = feature_maps[0].dtype, feature_maps[0].device
        key = (tuple(tuple(g) for g in grid_sizes), tuple(image_size), dtype, device)
        def make():                                     # the anchors depend on the shapes only
            per_level = []
            for (gh, gw), base in zip(grid_sizes, self.cell_anchors):
                sh, sw = image_size[0] // gh, image_size[1] // gw
                xs = torch.arange(0, gw, dtype=torch.int32, device=device) * sw
                ys = torch.arange(0, gh, dtype=torch.int32, device=device) * sh
                yy, xx = torch.meshgrid(ys, xs, indexing="ij")
                xx, yy = xx.reshape(-1), yy.reshape(-1)
                shifts = torch.stack((xx, yy, xx, yy), dim=1)
                per_level.append((shifts.view(-1, 1, 4) + base.to(device=device, dtype=dtype).view(1, -1, 4)).reshape(-1, 4))
            return torch.cat(per_level)
        all_anchors = self._cache.get(key, make, device)
        return [all_anchors for _ in image_list.image_sizes]
