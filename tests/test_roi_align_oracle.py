"""Pins of the RoIAlign restatement (oracle/roi_align_oracle.py; torchvision 0.13.1 is absent from /root/reference, so the
pins are closed-form known answers) and its agreement with the product's stock-torch stand-in on the CPU."""
import numpy as np
import pytest
import torch

from oracle import roi_align_oracle as RA


def test_constant_map_pools_to_the_constant():
    feat = np.full((3, 20, 30), 1.75, dtype=np.float32)
    out = RA.roi_align_single(feat, (3.3, 2.1, 17.9, 15.2), 1.0)
    assert out.shape == (3, 7, 7)
    np.testing.assert_allclose(out, 1.75, rtol=0, atol=2e-7)


def test_affine_map_pools_to_the_value_at_the_bin_centre():
    """bilinear interpolation reproduces an affine function exactly, and the mean of the 2x2 sample grid of a bin is the
    function at the bin centre: pooled[c, ph, pw] = f(y1 + (ph + .5) bh, x1 + (pw + .5) bw)   (RoI strictly inside)"""
    H, W = 40, 50
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    coef = [(0.25, -0.5, 3.0), (-1.0, 0.125, 0.0)]
    feat = np.stack([(a * yy + b * xx + c) for a, b, c in coef]).astype(np.float32)
    for scale, roi in ((1.0, (4.0, 6.0, 32.0, 27.0)), (0.5, (10.0, 8.0, 70.0, 66.0)), (0.25, (17.0, 33.0, 150.0, 120.0))):
        out = RA.roi_align_single(feat, roi, scale)
        x1, y1, x2, y2 = [v * scale for v in roi]
        bh, bw = max(y2 - y1, 1.0) / 7, max(x2 - x1, 1.0) / 7
        cy = y1 + (np.arange(7) + 0.5) * bh
        cx = x1 + (np.arange(7) + 0.5) * bw
        for ch, (a, b, c) in enumerate(coef):
            exp = a * cy[:, None] + b * cx[None, :] + c
            np.testing.assert_allclose(out[ch], exp, rtol=0, atol=2e-5)


def test_hand_computed_corner_cases():
    feat = np.arange(12, dtype=np.float32).reshape(1, 3, 4)               # f(y, x) = 4y + x
    # degenerate RoI: width/height clamp to 1 -> bins of 1/7; first bin's samples at 1/28 and 3/28 from the corner
    out = RA.roi_align_single(feat, (1.0, 1.0, 1.0, 1.0), 1.0)
    exp00 = 4 * (1 + 1 / 14) + (1 + 1 / 14)
    assert abs(float(out[0, 0, 0]) - exp00) < 1e-5
    # a RoI far outside the map: every sample has y > H or x > W -> zeros
    out = RA.roi_align_single(feat, (50.0, 50.0, 60.0, 60.0), 1.0)
    assert float(np.abs(out).max()) == 0.0
    # samples between -1 and 0 clamp to the border value; beyond the last row / column they read the last one
    out = RA.roi_align_single(feat, (-0.9, -0.9, -0.1, -0.1), 1.0)        # clamps to 1x1 starting at -0.9
    assert float(out[0, 0, 0]) == 0.0                                     # both samples of the first bin sit at y, x < 0 -> f(0, 0)
    out = RA.roi_align_single(feat, (3.2, 2.1, 3.9, 2.8), 1.0)            # y in (2.1, 3.1) > H-1 = 2, x in (3.2, 4.2) > W-1 = 3
    assert float(out[0, 0, 0]) == 11.0


def test_level_mapper_and_scales():
    assert RA.infer_scale((192, 384), (768, 1536)) == 0.25
    assert RA.infer_scale((24, 43), (768, 1365)) == 1.0 / 32
    boxes = np.array([[0, 0, 10, 10], [0, 0, 111.9, 111.9], [0, 0, 112.1, 112.1], [0, 0, 224, 224], [0, 0, 448, 448], [0, 0, 2000, 2000]],
                     dtype=np.float32)
    assert RA.map_levels(boxes, 2, 5).tolist() == [0, 0, 1, 2, 3, 3]


@pytest.mark.parametrize("seed", [0, 1])
def test_stock_stand_in_equals_the_restatement_on_cpu(seed):
    """stock/roi_align.py (vectorised torch; feeds the un-fused path and the bench inputs) against the scalar-order
    restatement: same levels, pooled values bit-identical on the CPU (both evaluate torchvision's operation order in IEEE fp32)"""
    from snn_automotive_object_detection_amd.stock.roi_align import MultiScaleRoIAlign
    g = torch.Generator().manual_seed(seed)
    sizes = [(48, 80), (24, 40), (12, 20), (6, 10)]
    feats = {str(i): torch.randn((2, 5, h, w), generator=g) for i, (h, w) in enumerate(sizes)}
    feats["pool"] = torch.randn((2, 5, 3, 5), generator=g)
    boxes = []
    for n in range(2):
        xy = torch.rand((40, 2), generator=g) * torch.tensor([300.0, 180.0])
        wh = torch.exp(torch.rand((40, 2), generator=g) * 5.0 + 1.0)
        b = torch.cat([xy, xy + wh], 1)
        b[0] = torch.tensor([-20.0, -10.0, 30.0, 25.0])
        b[1] = torch.tensor([315.0, 185.0, 400.0, 300.0])
        b[2] = torch.tensor([100.0, 100.0, 100.2, 100.1])
        boxes.append(b)
    shapes = [(192, 320), (192, 300)]
    stock = MultiScaleRoIAlign(["0", "1", "2", "3"], 7, 2)(feats, boxes, shapes)
    mine = RA.multiscale_roi_align(feats, boxes, shapes)
    assert stock.shape == mine.shape == (80, 5, 7, 7)
    assert torch.equal(stock, mine)


def test_golden_fixture_is_what_the_restatement_gives():
    from oracle import fixtures as FX
    spec = FX.ROI_SPECS["roialign_c8"]
    feats, boxes, shapes = FX.roi_inputs(spec)
    exp = FX.load_expected("roialign_c8")
    got = RA.multiscale_roi_align(feats, boxes, shapes)
    assert np.array_equal(got.numpy(), exp["pooled"])
