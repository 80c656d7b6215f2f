"""fc6 + LIF and fc7 + LIF of the detector head in ONE launch (k_gemm_bf16x3_pair, csrc/snn_bf16x3.h: fc7's row tiles wait for the
fc6 row tiles they read - faster_rcnn.py:498-501; SNN_DET_PAIR=1) against the default two-launch form: the same tiles run the same
arithmetic, so logits, deltas, spike counts and rate tensors must be bit-identical - for RoI counts that do and do not fill the
tiles, in spike-rate mode (whose fc6 window is one step longer than fc7's: the two layers' row tiles then do not coincide), on
the fused RoIAlign path and when called repeatedly (the tile counters are re-zeroed per call)."""
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


def _head(dev, D, Hd, K, T, seed):
    import snn_automotive_object_detection_amd as S
    torch.manual_seed(seed)
    d = S.FastRCNNPredictorSNNFull(D, Hd, K, T).to(dev)
    with torch.no_grad():
        d.fc7.weight.mul_(3.0)                     # fc7 spikes too
    return d


@pytest.mark.parametrize("R,C,Hd,K,T", [(2000, 256, 1024, 9, 12), (1, 32, 128, 5, 8), (45, 32, 128, 5, 4), (333, 64, 256, 11, 16),
                                        (1000, 32, 1024, 9, 24), (130, 32, 96, 3, 12)])
@pytest.mark.parametrize("rates", [False, True])
def test_pair_launch_equals_two_launches(gpu_device, monkeypatch, R, C, Hd, K, T, rates):
    d = _head(gpu_device, C * 49, Hd, K, T, R)
    d.spike_rates = rates
    x = torch.randn(R, C, 7, 7, device=gpu_device) * 2
    monkeypatch.setenv("SNN_DET_PAIR", "1")
    a = [t.clone() for t in d(x)]
    a2 = [t.clone() for t in d(x)]                 # second call on the same workspace
    cnt_a = [c.clone() for c in d.last_spike_counts] if rates else []
    monkeypatch.delenv("SNN_DET_PAIR")
    b = [t.clone() for t in d(x)]
    cnt_b = [c.clone() for c in d.last_spike_counts] if rates else []
    for p, q, r in zip(a, b, a2):
        assert torch.equal(p, q) and torch.equal(p, r)
    for p, q in zip(cnt_a, cnt_b):
        assert torch.equal(p, q)
    assert any(float(t.abs().max()) > 0 for t in a)
    torch.cuda.synchronize()


def test_pair_launch_under_a_captured_graph_and_two_streams(gpu_device, monkeypatch):
    """the counters are zeroed by a memset node of the same capture; two host threads on two streams use separate workspaces"""
    d = _head(gpu_device, 32 * 49, 256, 9, 12, 3)
    x = torch.randn(500, 32, 7, 7, device=gpu_device) * 2
    ref = [t.clone() for t in d(x)]
    monkeypatch.setenv("SNN_DET_PAIR", "1")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        d(x)                                        # warm the stream's workspace outside the capture
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            out = d(x)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    assert all(torch.equal(p, q) for p, q in zip(out, ref))


def test_pair_launch_on_the_fused_roialign_path(gpu_device, monkeypatch):
    import snn_automotive_object_detection_amd as S
    from tests.test_gpu_roialign import _setup
    d = _head(gpu_device, 32 * 49, 128, 9, 12, 9)
    pool, fm, boxes, shapes = _setup(gpu_device, R=700, C=32, seed=4)
    flist, scales, rois, lvl = pool.assign(fm, boxes, shapes)
    b = [t.clone() for t in d.forward_roialign(flist, scales, rois, lvl)]
    monkeypatch.setenv("SNN_DET_PAIR", "1")
    a = [t.clone() for t in d.forward_roialign(flist, scales, rois, lvl)]
    assert all(torch.equal(p, q) for p, q in zip(a, b))
