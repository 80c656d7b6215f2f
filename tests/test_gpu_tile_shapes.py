"""Tile heights of the T-in-tile kernels (csrc/snn_bf16x3.h: Gemm3Args.n_short; launchers: g3_pick_tile).  With half of the
row-waves one M-tile "short" a tile has 448 / 320 (8 x 1 wave grid) or 224 / 160 (4 x 2) rows instead of 512 / 384 / 256:
the same arithmetic on a different partition of the rows, so every output must be bit-identical to the full-wave shapes."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _dense_launches_only(monkeypatch):
    """these tests compare variants of the DENSE matrix-core launches bit for bit; the structured-sparse launches of round 4
    (csrc/snn_sparse.h: another fp32 summation order, their own tests in tests/test_gpu_sparse.py) and fc6's permuted reduction
    order (word-major planes only) are switched off"""
    monkeypatch.setenv("SNN_SPARSE", "0")
    monkeypatch.setenv("SNN_FC6_PERM", "0")


def _short_vs_full(monkeypatch, fn):
    monkeypatch.setenv("SNN_BF16X3_SHORT", "0")
    full = fn()
    monkeypatch.setenv("SNN_BF16X3_SHORT", "1")
    short = fn()
    monkeypatch.delenv("SNN_BF16X3_SHORT")
    auto = fn()
    for a, b, c in zip(full, short, auto):
        assert torch.equal(a, b) and torch.equal(a, c)
    return full


@pytest.mark.parametrize("wn", ["1", "2"])
@pytest.mark.parametrize("mt", ["4", pytest.param("3", marks=pytest.mark.sweep), "2"])
@pytest.mark.parametrize("R,Hd,K,T", [(300, 128, 9, 12), (2000, 160, 9, 12), (37, 96, 5, 8), (513, 256, 11, 16), (64, 64, 3, 5), (90, 64, 3, 24)])
def test_det_head_short_waves(gpu_device, monkeypatch, wn, mt, R, Hd, K, T):
    import snn_automotive_object_detection_amd as S
    torch.manual_seed(R + T)
    m = S.FastRCNNPredictorSNNFull(32 * 49, Hd, K, T).to(gpu_device)
    with torch.no_grad():
        m.fc7.weight.mul_(3.0)
    x = torch.randn(R, 32, 7, 7, device=gpu_device) * 1.5
    monkeypatch.setenv("SNN_BF16X3_WN", wn)
    monkeypatch.setenv("SNN_BF16X3_MT", mt)

    def run():
        m.spike_rates = False
        c, b = m(x)
        m.spike_rates = True
        rates = m(x)
        return [c.clone(), b.clone()] + [r.clone() for r in rates] + [t.clone() for t in m.last_spike_counts]
    out = _short_vs_full(monkeypatch, run)
    if T >= 8:
        assert int(out[-1].sum()) > 0


@pytest.mark.parametrize("wn", ["1", "2"])
@pytest.mark.parametrize("mt", ["4", pytest.param("3", marks=pytest.mark.sweep), "2"])
@pytest.mark.parametrize("C,T,shapes", [(256, 8, [(2, 48, 96), (2, 24, 48), (2, 12, 24), (2, 6, 12), (2, 3, 6)]),
                                        (96, 12, [(1, 9, 14), (3, 5, 7), (1, 1, 1)]), (64, 4, [(2, 7, 33)]), (64, 24, [(1, 11, 13)])])
def test_rpn_head_short_waves(gpu_device, monkeypatch, wn, mt, C, T, shapes):
    import snn_automotive_object_detection_amd as S
    torch.manual_seed(C + T)
    m = S.RPNHeadSNN(C, 3, T).to(gpu_device)
    with torch.no_grad():
        m.shared_conv.weight.mul_(5.0)
    m.spike_rates = True
    feats = [torch.randn(n, C, h, w, device=gpu_device) * 1.5 for n, h, w in shapes]
    monkeypatch.setenv("SNN_BF16X3_WN", wn)
    monkeypatch.setenv("SNN_BF16X3_MT", mt)

    def run():
        lg, bb, rates = m(feats)
        return [x.clone() for x in lg + bb + list(rates)] + [m.last_spike_counts.clone()]
    out = _short_vs_full(monkeypatch, run)
    assert int(out[-1].sum()) > 0


def test_default_detector_launch_uses_the_short_shape(gpu_device, monkeypatch, capfd):
    """2000 RoIs x 10 current steps x 16 column blocks: 448-row tiles (46 x 16 = 736 work-groups, 3 rounds of 7/8-size tiles per
    CU) instead of 512-row tiles (640 work-groups: 3 rounds of full-size tiles)"""
    import snn_automotive_object_detection_amd as S
    monkeypatch.setenv("SNN_DEBUG_OCC", "1")
    m = S.FastRCNNPredictorSNNFull(32 * 49, 1024, 9, 12).to(gpu_device)
    m(torch.randn(2000, 32, 7, 7, device=gpu_device))
    torch.cuda.synchronize()
    err = capfd.readouterr().err
    assert "short 4" in err, err
