"""ORACLE — test infrastructure only (tests/, bench.py's cpu_baseline leg).  Never imported by the product path.

CPU restatement of the box pooling in front of the detector head:

    box_features = self.box_roi_pool(features, proposals, image_shapes)     /root/reference/roi_heads.py:1217
    box_roi_pool = MultiScaleRoIAlign(['0','1','2','3'], output_size=7, sampling_ratio=2)   /root/reference/model.py:118

``MultiScaleRoIAlign`` / ``roi_align`` live in torchvision 0.13.1, a third-party dependency that is ABSENT from
/root/reference (pinned in prose only, README.md:13-17) and not installed here: "parity unpinned" by the reference.  What
follows restates the published algorithm of that release:

* torchvision/ops/poolers.py: ``_infer_scale`` (scale = 2^round(log2(feature height / image height))), ``LevelMapper``
  (level = floor(4 + log2(sqrt(area) / 224) + 1e-6) clamped to [k_min, k_max]), ``_multiscale_roi_align`` (RoIs grouped by
  level, one roi_align per level, results scattered back);
* torchvision/csrc/ops/cpu/roi_align_kernel.cpp + roi_align_common.h (``roi_align_forward_kernel_impl``,
  ``pre_calc_for_bilinear_interpolate``), aligned=False: every fp32 operation in the order the C++ evaluates it
  (no fused multiply-add: the x86-64 wheels are built without FMA).

Independent of ``snn_automotive_object_detection_amd/stock/roi_align.py`` (vectorised torch, the product's stand-in) and of
the HIP kernel ``k_roi_align_encode`` - the three are compared in tests/test_roi_align_oracle.py (CPU) and
tests/test_gpu_roialign.py.  Pins: closed-form known answers (constant map, affine map: bilinear interpolation is exact on
affine functions, hand-computed corner cases) in tests/test_roi_align_oracle.py.
"""
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

F32 = np.float32


def infer_scale(feature_hw: Tuple[int, int], image_hw: Tuple[int, int]) -> float:
    """poolers.py::_infer_scale: the HEIGHT ratio decides (possible_scales[0])"""
    approx = float(feature_hw[0]) / float(image_hw[0])
    return 2.0 ** float(torch.tensor(approx).log2().round())


def map_levels(boxes: np.ndarray, k_min: int, k_max: int, canonical_scale: float = 224.0, canonical_level: float = 4.0,
               eps: float = 1e-6) -> np.ndarray:
    """poolers.py::LevelMapper.__call__ (fp32 torch ops on the CPU, as the reference evaluates them)"""
    b = torch.from_numpy(np.ascontiguousarray(boxes, dtype=F32))
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])                     # box_area
    s = torch.sqrt(area)
    lvl = torch.floor(canonical_level + torch.log2(s / canonical_scale) + torch.tensor(eps, dtype=s.dtype))
    lvl = torch.clamp(lvl, min=k_min, max=k_max)
    return (lvl.to(torch.int64) - k_min).numpy()


def roi_align_single(feat: np.ndarray, roi: Sequence[float], spatial_scale: float, pooled: int = 7,
                     sampling_ratio: int = 2) -> np.ndarray:
    """one RoI on one image's feature map: feat [C,H,W] fp32, roi = (x1,y1,x2,y2) in image coordinates -> [C,pooled,pooled].
    roi_align_kernel.cpp::roi_align_forward_kernel_impl with aligned=False, vectorised over channels and samples only
    (each element sees exactly the scalar code's operation sequence)."""
    assert sampling_ratio > 0, "the reference uses sampling_ratio=2 (model.py:118)"
    C, H, W = feat.shape
    sc = F32(spatial_scale)
    offset = F32(0.0)                                                   # aligned=False
    roi_start_w = F32(F32(roi[0]) * sc) - offset
    roi_start_h = F32(F32(roi[1]) * sc) - offset
    roi_end_w = F32(F32(roi[2]) * sc) - offset
    roi_end_h = F32(F32(roi[3]) * sc) - offset
    roi_width = max(F32(roi_end_w - roi_start_w), F32(1.0))              # malformed RoIs are forced to 1x1
    roi_height = max(F32(roi_end_h - roi_start_h), F32(1.0))
    bin_h = F32(roi_height / F32(pooled))
    bin_w = F32(roi_width / F32(pooled))
    g = sampling_ratio
    count = F32(max(g * g, 1))
    ph = np.repeat(np.arange(pooled), g).astype(F32)                    # sample row i = ph*g + iy
    iy = np.tile(np.arange(g), pooled).astype(F32)
    # yy = roi_start_h + ph * bin_size_h + (iy + .5f) * bin_size_h / grid      (left to right, fp32)
    yy = (roi_start_h + ph * bin_h).astype(F32) + (((iy + F32(0.5)) * bin_h).astype(F32) / F32(g)).astype(F32)
    xx = (roi_start_w + ph * bin_w).astype(F32) + (((iy + F32(0.5)) * bin_w).astype(F32) / F32(g)).astype(F32)
    y = np.broadcast_to(yy[:, None], (pooled * g, pooled * g)).astype(F32)
    x = np.broadcast_to(xx[None, :], (pooled * g, pooled * g)).astype(F32)
    empty = (y < -1.0) | (y > H) | (x < -1.0) | (x > W)
    y = np.where(y <= 0, F32(0), y)
    x = np.where(x <= 0, F32(0), x)
    y_low = y.astype(np.int64)                                           # (int)y: truncation, y >= 0
    x_low = x.astype(np.int64)
    y_edge, x_edge = y_low >= H - 1, x_low >= W - 1
    y_high = np.where(y_edge, H - 1, y_low + 1)
    x_high = np.where(x_edge, W - 1, x_low + 1)
    y_low = np.where(y_edge, H - 1, y_low)
    x_low = np.where(x_edge, W - 1, x_low)
    y = np.where(y_edge, y_low.astype(F32), y)
    x = np.where(x_edge, x_low.astype(F32), x)
    ly = (y - y_low.astype(F32)).astype(F32)
    lx = (x - x_low.astype(F32)).astype(F32)
    hy = (F32(1.0) - ly).astype(F32)
    hx = (F32(1.0) - lx).astype(F32)
    w1, w2, w3, w4 = (hy * hx).astype(F32), (hy * lx).astype(F32), (ly * hx).astype(F32), (ly * lx).astype(F32)
    for w in (w1, w2, w3, w4):
        w[empty] = 0
    y_low = np.where(empty, 0, y_low); x_low = np.where(empty, 0, x_low)
    y_high = np.where(empty, 0, y_high); x_high = np.where(empty, 0, x_high)
    v1, v2 = feat[:, y_low, x_low], feat[:, y_low, x_high]               # [C, PG, PG]
    v3, v4 = feat[:, y_high, x_low], feat[:, y_high, x_high]
    # output_val += w1*v1 + w2*v2 + w3*v3 + w4*v4     (one sample)
    s = (((w1 * v1).astype(F32) + (w2 * v2).astype(F32)).astype(F32) + (w3 * v3).astype(F32)).astype(F32)
    s = (s + (w4 * v4).astype(F32)).astype(F32)
    s = s.reshape(C, pooled, g, pooled, g)
    out = np.zeros((C, pooled, pooled), dtype=F32)                       # output_val = 0, then += in (iy, ix) order
    for a in range(g):
        for b in range(g):
            out = (out + s[:, :, a, :, b]).astype(F32)
    return (out / count).astype(F32)


def multiscale_roi_align(features: Dict[str, torch.Tensor], boxes: List[torch.Tensor], image_shapes: List[Tuple[int, int]],
                         featmap_names: Sequence[str] = ("0", "1", "2", "3"), output_size: int = 7,
                         sampling_ratio: int = 2) -> torch.Tensor:
    """poolers.py::MultiScaleRoIAlign.forward: features {name: [N,C,H,W]}, boxes per image [k_i,4] -> [sum k_i, C, 7, 7]"""
    feats = [features[k].detach().cpu().numpy().astype(F32) for k in features if k in featmap_names]
    max_h = max(int(s[0]) for s in image_shapes)
    max_w = max(int(s[1]) for s in image_shapes)
    scales = [infer_scale(f.shape[-2:], (max_h, max_w)) for f in feats]
    allb = np.concatenate([b.detach().cpu().numpy().astype(F32).reshape(-1, 4) for b in boxes], 0)
    img = np.concatenate([np.full((int(b.shape[0]),), i, dtype=np.int64) for i, b in enumerate(boxes)], 0)
    C = feats[0].shape[1]
    out = np.zeros((allb.shape[0], C, output_size, output_size), dtype=F32)
    if len(feats) == 1:
        lvl = np.zeros((allb.shape[0],), dtype=np.int64)
    else:
        k_min = int(-float(torch.log2(torch.tensor(scales[0], dtype=torch.float32))))
        k_max = int(-float(torch.log2(torch.tensor(scales[-1], dtype=torch.float32))))
        lvl = map_levels(allb, k_min, k_max)
    for r in range(allb.shape[0]):
        out[r] = roi_align_single(feats[lvl[r]][img[r]], allb[r], scales[lvl[r]], output_size, sampling_ratio)
    return torch.from_numpy(out)
