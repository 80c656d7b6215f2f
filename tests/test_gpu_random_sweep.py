"""Seeded random sweep of both heads against the oracle run on the spot: ragged pyramids, channel counts that are not
multiples of 32 or 128, 1-24 time steps, anchor / class counts, RoI counts around the tile sizes.  Complements the fixed
fixtures: every default layout choice of the bf16x3 path (tile shapes, word-major planes, spike planes in blocks of four
words where C = 256, XCD-aware block order) is crossed here by shapes nobody picked by hand.  Tolerance as everywhere:
1e-4 on the outputs, a small budget of positions / RoIs whose hidden spikes flipped (fp32 summation order)."""
import numpy as np
import pytest
import torch

from oracle import snn_oracle as OR
from tests._util import flip_budget

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _rpn_case(seed):
    r = np.random.default_rng(seed)
    C = int(r.choice([16, 40, 64, 96, 128, 200, 256]))
    A = int(r.choice([1, 3, 5]))
    T = int(r.choice([1, 3, 4, 8, 12, 16, 24]))
    n_levels = int(r.integers(1, 4))
    N = int(r.integers(1, 4))
    shapes = [(N, int(r.integers(1, 15)), int(r.integers(1, 19))) for _ in range(n_levels)]
    return C, A, T, shapes


@pytest.mark.parametrize("seed", range(16))
@pytest.mark.parametrize("precision", ["bf16x3", pytest.param("mxfp6", marks=pytest.mark.sweep)])
def test_rpn_head_random_shapes_vs_oracle(gpu_device, seed, precision):
    import snn_automotive_object_detection_amd as S
    C, A, T, shapes = _rpn_case(seed)
    g = torch.Generator().manual_seed(1000 + seed)
    m = S.RPNHeadSNN(C, A, T)
    with torch.no_grad():
        m.shared_conv.weight.mul_(4.0)                          # let the shared LIF fire
    feats = [torch.randn((n, C, h, w), generator=g) * 1.5 for n, h, w in shapes]
    with torch.no_grad():
        o_l, o_b = OR.rpn_head_forward(feats, m.shared_conv.weight, m.conv_cls.weight, m.conv_bbox.weight, T)
    m = m.to(gpu_device)
    m.precision = precision                                     # mxfp6 needs C % 128 == 0 and falls back to bf16x3 otherwise
    l, b = m([f.to(gpu_device) for f in feats])
    off = 0
    P = 0
    for lv in range(len(shapes)):
        d = torch.maximum((l[lv].cpu() - o_l[lv]).abs().amax(1), (b[lv].cpu() - o_b[lv]).abs().amax(1))
        off += int((d > TOL).sum())
        P += d.numel()
        assert float(d.max()) < 0.2, (seed, lv, float(d.max()))
    assert off <= flip_budget(P, C, T, "rpn_in_situ", precision), (seed, C, A, T, shapes, off)


def _det_case(seed):
    r = np.random.default_rng(100 + seed)
    Cc = int(r.choice([2, 8, 13, 32, 64]))
    Hd = int(r.choice([32, 40, 64, 96, 128, 192]))
    K = int(r.choice([2, 5, 9, 11]))
    T = int(r.choice([1, 4, 8, 12, 16, 24]))
    R = int(r.choice([1, 7, 20, 21, 22, 41, 42, 43, 85, 130, 257]))
    return Cc, Hd, K, T, R


@pytest.mark.parametrize("seed", range(16))
def test_det_head_random_shapes_vs_oracle(gpu_device, seed):
    import snn_automotive_object_detection_amd as S
    Cc, Hd, K, T, R = _det_case(seed)
    g = torch.Generator().manual_seed(2000 + seed)
    m = S.FastRCNNPredictorSNNFull(Cc * 49, Hd, K, T)
    x = torch.randn((R, Cc, 7, 7), generator=g) * 1.5
    with torch.no_grad():
        o_c, o_b = OR.det_head_forward(x, m.fc6.weight, m.fc7.weight, m.cls_score.weight, m.bbox_pred.weight, T)
    m = m.to(gpu_device)
    c, b = m(x.to(gpu_device))
    d = torch.maximum((c.cpu() - o_c).abs().amax(1), (b.cpu() - o_b).abs().amax(1))
    assert int((d > TOL).sum()) <= flip_budget(R, Hd, T) and float(d.max()) < 0.5, (seed, Cc, Hd, K, T, R, int((d > TOL).sum()), float(d.max()))
