"""MultiScaleRoIAlign(['0','1','2','3'], 7, 2) in plain torch (model.py:118, roi_heads.py:1217):
FPN level mapper (canonical 224 / level 4, k in [2,5]) + roi_align(aligned=False, sampling_ratio=2)."""
from typing import Dict, List, Tuple

import torch
from torch import nn, Tensor


def _bilinear(feat: Tensor, y: Tensor, x: Tensor) -> Tensor:
    """feat [C,H,W]; y, x [...]: torchvision roi_align's bilinear_interpolate -> [C, ...]"""
    C, H, W = feat.shape
    outside = (y < -1.0) | (y > H) | (x < -1.0) | (x > W)
    y = y.clamp(min=0)
    x = x.clamp(min=0)
    y_low = y.floor().to(torch.int64)
    x_low = x.floor().to(torch.int64)
    y_edge = y_low >= H - 1
    x_edge = x_low >= W - 1
    y_high = torch.where(y_edge, torch.full_like(y_low, H - 1), y_low + 1)
    x_high = torch.where(x_edge, torch.full_like(x_low, W - 1), x_low + 1)
    y_low = torch.where(y_edge, torch.full_like(y_low, H - 1), y_low)
    x_low = torch.where(x_edge, torch.full_like(x_low, W - 1), x_low)
    y = torch.where(y_edge, y_low.to(y.dtype), y)
    x = torch.where(x_edge, x_low.to(x.dtype), x)
    ly, lx = y - y_low.to(y.dtype), x - x_low.to(x.dtype)
    hy, hx = 1.0 - ly, 1.0 - lx
    flat = feat.reshape(C, H * W)
    def g(yy, xx):
        return flat[:, (yy * W + xx).reshape(-1)].reshape((C,) + tuple(yy.shape))
    val = hy * hx * g(y_low, x_low) + hy * lx * g(y_low, x_high) + ly * hx * g(y_high, x_low) + ly * lx * g(y_high, x_high)
    return torch.where(outside, torch.zeros_like(val), val)


def roi_align(feat: Tensor, rois: Tensor, spatial_scale: float, output_size: int = 7, sampling_ratio: int = 2,
              chunk: int = 256) -> Tensor:
    """feat [N,C,H,W]; rois [K,5] (batch index, x1,y1,x2,y2) -> [K,C,output_size,output_size] (aligned=False)"""
    K = rois.shape[0]
    C = feat.shape[1]
    out = feat.new_zeros((K, C, output_size, output_size))
    if K == 0:
        return out
    S, P = sampling_ratio, output_size
    idx = torch.arange(P * S, device=feat.device, dtype=feat.dtype)
    g_p = torch.div(idx, S, rounding_mode="floor")                                      # ph of sample row / column i
    g_i = (idx % S) + 0.5                                                               # iy + .5
    for n in range(feat.shape[0]):
        idx = torch.where(rois[:, 0] == n)[0]
        for s in range(0, idx.numel(), chunk):
            sel = idx[s:s + chunk]
            r = rois[sel]
            x1, y1 = r[:, 1] * spatial_scale, r[:, 2] * spatial_scale
            rw = (r[:, 3] * spatial_scale - x1).clamp(min=1.0)
            rh = (r[:, 4] * spatial_scale - y1).clamp(min=1.0)
            bh, bw = (rh / P)[:, None], (rw / P)[:, None]
            # torchvision's operation order (roi_align_common.h): start + ph * bin + (iy + .5) * bin / S, left to right
            ys = (y1[:, None] + g_p[None, :] * bh) + (g_i[None, :] * bh) / S            # [k, P*S]
            xs = (x1[:, None] + g_p[None, :] * bw) + (g_i[None, :] * bw) / S
            yy = ys[:, :, None].expand(-1, -1, P * S)
            xx = xs[:, None, :].expand(-1, P * S, -1)
            v = _bilinear(feat[n], yy, xx)                                              # [C, k, PS, PS]
            v = v.reshape(C, sel.numel(), P, S, P, S)
            acc = torch.zeros_like(v[:, :, :, 0, :, 0])
            for a in range(S):                                                          # output_val += sample, (iy, ix) order
                for b in range(S):
                    acc = acc + v[:, :, :, a, :, b]
            v = acc / float(S * S)
            out[sel] = v.permute(1, 0, 2, 3)
    return out


class MultiScaleRoIAlign(nn.Module):
    def __init__(self, featmap_names: List[str], output_size, sampling_ratio: int, canonical_scale: int = 224,
                 canonical_level: int = 4):
        super().__init__()
        if isinstance(output_size, int):
            output_size = (output_size, output_size)
        self.featmap_names = featmap_names
        self.output_size = tuple(output_size)
        self.sampling_ratio = sampling_ratio
        self.canonical_scale, self.canonical_level = canonical_scale, canonical_level

    def assign(self, x: Dict[str, Tensor], boxes: List[Tensor], image_shapes: List[Tuple[int, int]]):
        """-> (feature maps used, their scales, rois [K,5] = (image index, x1,y1,x2,y2), FPN level index per RoI)"""
        feats = [v for k, v in x.items() if k in self.featmap_names]
        max_h = max(s[0] for s in image_shapes)
        scales = []
        for f in feats:                                   # scale = 2^round(log2(feature / image))
            approx = float(f.shape[-2]) / float(max_h)
            scales.append(2.0 ** float(torch.tensor(approx).log2().round()))
        ids = torch.cat([torch.full((b.shape[0], 1), i, dtype=b.dtype, device=b.device) for i, b in enumerate(boxes)])
        allb = torch.cat(list(boxes), dim=0)
        rois = torch.cat([ids, allb], dim=1)
        if len(feats) == 1:
            return feats, scales, rois, torch.zeros((rois.shape[0],), dtype=torch.int64, device=rois.device)
        k_min = -int(round(float(torch.log2(torch.tensor(scales[0])))))
        k_max = -int(round(float(torch.log2(torch.tensor(scales[-1])))))
        s = torch.sqrt((allb[:, 2] - allb[:, 0]) * (allb[:, 3] - allb[:, 1]))
        lvl = torch.floor(self.canonical_level + torch.log2(s / self.canonical_scale) + torch.tensor(1e-6, dtype=s.dtype))
        lvl = (torch.clamp(lvl, min=k_min, max=k_max).to(torch.int64) - k_min)
        return feats, scales, rois, lvl

    def forward(self, x: Dict[str, Tensor], boxes: List[Tensor], image_shapes: List[Tuple[int, int]]) -> Tensor:
        feats, scales, rois, lvl = self.assign(x, boxes, image_shapes)
        C = feats[0].shape[1]
        out = feats[0].new_zeros((rois.shape[0], C) + self.output_size)
        for level, (f, sc) in enumerate(zip(feats, scales)):
            sel = torch.where(lvl == level)[0]
            if sel.numel():
                out[sel] = roi_align(f, rois[sel], sc, self.output_size[0], self.sampling_ratio).to(out.dtype)
        return out
