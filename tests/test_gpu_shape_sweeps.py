"""Shape sweeps of both heads against the oracle (rpn.py:84-121, faster_rcnn.py:470-516) at default knobs: every channel count,
anchor / class count, hidden width and row remainder selects tiles, wave grids, column blocks, resident / streamed / split head kernels
and sparse / dense launches on the host side (csrc/snn_kernels.hip) - a defect can sit in one combination (cf. tests/test_gpu_every_T.py).
Small inputs: the oracle takes well under a second per case.  The default `-m gpu` run takes the values on either side of every tile / block
boundary; the values in between are marked `sweep` (`-m "gpu and sweep"`, tests/conftest.py)."""
import pytest
import torch

from oracle import snn_oracle as OR
from tests._util import flip_budget

pytestmark = pytest.mark.gpu


def _rpn_case(dev, C, A, T, shapes, N, seed, precision="bf16x3"):
    import snn_automotive_object_detection_amd as S
    g = torch.Generator().manual_seed(seed)
    feats = [torch.randn(N, C, h, w, generator=g) * 1.7 for h, w in shapes]
    torch.manual_seed(seed)
    m = S.RPNHeadSNN(C, A, T).to(dev)
    m.precision = precision
    with torch.no_grad():
        m.shared_conv.weight.mul_(4.0)
    lg, bb = m([f.to(dev) for f in feats])
    o_l, o_b = OR.rpn_head_forward(feats, m.shared_conv.weight.detach().cpu(), m.conv_cls.weight.detach().cpu(),
                                   m.conv_bbox.weight.detach().cpu(), T)
    bad = 0
    for l in range(len(shapes)):
        assert lg[l].shape == o_l[l].shape and bb[l].shape == o_b[l].shape
        d = torch.maximum((lg[l].cpu() - o_l[l]).abs().amax(1), (bb[l].cpu() - o_b[l]).abs().amax(1))
        bad += int((d > 1e-4).sum())
    pos = sum(N * h * w for h, w in shapes)
    assert bad <= flip_budget(pos, C, T, "rpn_randn", precision), (C, A, T, N, bad)
    return bad


@pytest.mark.parametrize("C", [3, pytest.param(20, marks=pytest.mark.sweep), 32, 64, pytest.param(96, marks=pytest.mark.sweep), 100, pytest.param(128, marks=pytest.mark.sweep), pytest.param(160, marks=pytest.mark.sweep), 192, pytest.param(224, marks=pytest.mark.sweep), 256, 320, pytest.param(384, marks=pytest.mark.sweep), 512])
def test_rpn_head_channel_counts(gpu_device, C):
    total = 0
    for T, A in [(8, 3), (6, 5), (12, 1)]:
        total += _rpn_case(gpu_device, C, A, T, [(13, 17), (6, 7), (2, 1)], 2, C + T)
    assert total <= 3


@pytest.mark.parametrize("A", [1, pytest.param(2, marks=pytest.mark.sweep), 3, 4, pytest.param(5, marks=pytest.mark.sweep), pytest.param(6, marks=pytest.mark.sweep), 9, pytest.param(12, marks=pytest.mark.sweep), 13, pytest.param(15, marks=pytest.mark.sweep), 16])
def test_rpn_head_anchor_counts(gpu_device, A):
    """5 A outputs per position: 16-column head tiles 1 .. 4, then a second launch for the columns beyond 64"""
    total = 0
    for C in (256, 64):
        total += _rpn_case(gpu_device, C, A, 8, [(9, 11), (3, 5)], 1, 7 * A + C)
    assert total <= 2


@pytest.mark.parametrize("N,shapes", [(1, [(1, 1)]), (3, [(1, 1), (1, 1)]), (1, [(16, 16)]), (1, [(15, 17)]), (2, [(8, 8), (8, 8), (8, 8), (8, 8), (8, 8)]),
                                      (5, [(7, 3)]), (1, [(1, 63)]), (1, [(65, 1)]), (4, [(4, 4), (2, 2), (1, 1)]), (1, [(33, 31), (1, 1)])])
def test_rpn_head_level_shapes(gpu_device, N, shapes):
    """row remainders against every tile size: positions = 1, 3, 256, 255, 640, 105, 63, 65, 84, 1024"""
    total = 0
    for C, T in [(256, 8), (256, 10), (64, 5), (128, 16)]:
        total += _rpn_case(gpu_device, C, 3, T, shapes, N, N + 10 * len(shapes) + C + T)
    assert total <= 3


def _det_case(dev, R, C, Hd, K, T, seed, precision="bf16x3"):
    import snn_automotive_object_detection_amd as S
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(R, C, 7, 7, generator=g) * 2
    torch.manual_seed(seed)
    d = S.FastRCNNPredictorSNNFull(C * 49, Hd, K, T).to(dev)
    d.precision = precision
    with torch.no_grad():
        d.fc7.weight.mul_(3.0)
    c, b = d(x.to(dev))
    o_c, o_d = OR.det_head_forward(x, d.fc6.weight.detach().cpu(), d.fc7.weight.detach().cpu(), d.cls_score.weight.detach().cpu(),
                                   d.bbox_pred.weight.detach().cpu(), T)
    assert c.shape == o_c.shape and b.shape == o_d.shape
    off = ((c.cpu() - o_c).abs().amax(1) > 1e-4) | ((b.cpu() - o_d).abs().amax(1) > 1e-4)
    assert int(off.sum()) <= flip_budget(R, 2 * Hd, T, "det", precision), (R, C, Hd, K, T, int(off.sum()))
    return int(off.sum())


@pytest.mark.parametrize("Hd", [8, pytest.param(32, marks=pytest.mark.sweep), 40, 64, pytest.param(96, marks=pytest.mark.sweep), 100, pytest.param(128, marks=pytest.mark.sweep), pytest.param(192, marks=pytest.mark.sweep), 256, pytest.param(320, marks=pytest.mark.sweep), pytest.param(512, marks=pytest.mark.sweep), 1024])
def test_det_head_hidden_widths(gpu_device, Hd):
    total = 0
    for C, K, T in [(32, 9, 12), (8, 5, 8), (64, 2, 6)]:
        total += _det_case(gpu_device, 41, C, Hd, K, T, Hd + C)
    assert total <= 3


@pytest.mark.parametrize("C", [1, pytest.param(3, marks=pytest.mark.sweep), 8, pytest.param(16, marks=pytest.mark.sweep), 32, 40, 64, pytest.param(96, marks=pytest.mark.sweep), 128, 256, 352, 512])
def test_det_head_channel_counts(gpu_device, C):
    """D = 49 C: multiples of 32 channels take the bin-major fc6 order (k_permute_planes in passes of 8 channel blocks: any channel count -
    ADVICE r4: C >= 352 used to be refused) and, where D / 32 is even, the structured-sparse launch; the others the reference's order"""
    total = 0
    for Hd, K, T in [(128, 9, 12), (64, 3, 5)]:
        total += _det_case(gpu_device, 29, C, Hd, K, T, 3 * C + Hd)
    assert total <= 2


@pytest.mark.parametrize("K", [2, pytest.param(3, marks=pytest.mark.sweep), pytest.param(4, marks=pytest.mark.sweep), pytest.param(7, marks=pytest.mark.sweep), 9, 11, 13, pytest.param(14, marks=pytest.mark.sweep), 16, pytest.param(21, marks=pytest.mark.sweep), 24, pytest.param(52, marks=pytest.mark.sweep), 91])
def test_det_head_class_counts(gpu_device, K):
    """5 K outputs per RoI: head tiles of 16 columns 1 .. 4; beyond 64 outputs (K >= 13) one launch per block of 64 columns (the
    reference's configs: cityscapes 9, bdd 11, idd 16, pascal 24, coco 91)"""
    total = 0
    for C, Hd in [(32, 1024), (32, 128)]:
        total += _det_case(gpu_device, 23, C, Hd, K, 12, K + Hd)
    assert total <= 2


@pytest.mark.parametrize("R", [1, pytest.param(2, marks=pytest.mark.sweep), 15, 16, 17, pytest.param(31, marks=pytest.mark.sweep), pytest.param(32, marks=pytest.mark.sweep), 33, pytest.param(47, marks=pytest.mark.sweep), pytest.param(48, marks=pytest.mark.sweep), pytest.param(49, marks=pytest.mark.sweep), pytest.param(63, marks=pytest.mark.sweep), 64, 65, pytest.param(127, marks=pytest.mark.sweep), pytest.param(129, marks=pytest.mark.sweep), 257])
def test_det_head_row_remainders(gpu_device, R):
    total = 0
    for C, Hd, T in [(32, 256, 12), (64, 128, 7)]:
        total += _det_case(gpu_device, R, C, Hd, 9, T, R + Hd)
    assert total <= 2


def test_no_rois_and_no_positions(gpu_device):
    """R = 0 (an image batch without proposals): empty outputs of the right shapes, as the reference's modules give"""
    import snn_automotive_object_detection_amd as S
    d = S.FastRCNNPredictorSNNFull(32 * 49, 128, 9, 12).to(gpu_device)
    c, b = d(torch.zeros(0, 32, 7, 7, device=gpu_device))
    assert tuple(c.shape) == (0, 9) and tuple(b.shape) == (0, 36)
