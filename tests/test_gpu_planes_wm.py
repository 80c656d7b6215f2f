"""Word-major spike planes ([T][word][row]: csrc/snn_bf16x3.h Gemm3Args.wm, snn_encode.h K1a / K1b'' / K1c'; the default inside
the bf16x3 detector head, SNN_PLANES=wm also for the RPN convolution) against the row-major planes of the stage-level ABI
(SNN_PLANES=rm): the layout changes which bytes a chunk fetches, not one arithmetic operation - outputs, spike counts and rate
tensors must be bit-identical."""
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


def _both(monkeypatch, fn):
    monkeypatch.setenv("SNN_PLANES", "rm")
    rm = fn()
    monkeypatch.setenv("SNN_PLANES", "wm")
    wm = fn()
    monkeypatch.delenv("SNN_PLANES")
    default = fn()                          # linear layers word-major, convolution row-major
    for a, b in zip(rm, default):
        assert torch.equal(a, b)
    return rm, wm


@pytest.mark.parametrize("C,T,shapes", [
    (256, 8, [(2, 48, 96), (2, 24, 48), (2, 12, 24), (2, 6, 12), (2, 3, 6)]),      # the Cityscapes pyramid at 1/4 size
    (96, 12, [(1, 9, 14), (3, 5, 7), (1, 1, 1)]),                                  # ragged: 3 channel words, 1x1 level
    (40, 3, [(2, 7, 33)]),                                                         # padded channel word, rows wider than a tile
    (64, 24, [(1, 11, 13)]),                                                       # T = 24: 10 positions per tile
    (320, 16, [(1, 10, 10), (2, 4, 5)]),                                           # 10 channel words
])
@pytest.mark.parametrize("tile", [("2", "4"), ("2", "2"), ("1", "4")])
def test_rpn_head_word_major_equals_row_major(gpu_device, monkeypatch, C, T, shapes, tile):
    import snn_automotive_object_detection_amd as S
    torch.manual_seed(C + T)
    m = S.RPNHeadSNN(C, 3, T).to(gpu_device)
    with torch.no_grad():
        m.shared_conv.weight.mul_(5.0)                          # let the shared LIF fire
    m.spike_rates = True
    feats = [torch.randn(n, C, h, w, device=gpu_device) * 1.5 for n, h, w in shapes]
    monkeypatch.setenv("SNN_BF16X3_WN", tile[0])
    monkeypatch.setenv("SNN_BF16X3_MT", tile[1])

    def run():
        lg, bb, rates = m(feats)
        return [x.clone() for x in lg + bb + list(rates)] + [m.last_spike_counts.clone()]
    rm, wm = _both(monkeypatch, run)
    assert int(rm[-1].sum()) > 0
    for a, b in zip(rm, wm):
        assert torch.equal(a, b)


@pytest.mark.parametrize("R,D_ch,Hd,K,T", [(300, 32, 128, 9, 12), (37, 64, 96, 5, 8), (1, 32, 64, 3, 24), (513, 32, 256, 11, 4),
                                          (2000, 32, 160, 9, 12), (45, 96, 40, 2, 16),
                                          (300, 16, 128, 9, 12)])      # D = 784, not a multiple of 32: row-major planes either way
@pytest.mark.parametrize("tile", [("2", "4"), ("2", "3"), ("1", "4")])
def test_det_head_word_major_equals_row_major(gpu_device, monkeypatch, R, D_ch, Hd, K, T, tile):
    import snn_automotive_object_detection_amd as S
    torch.manual_seed(R + T)
    m = S.FastRCNNPredictorSNNFull(D_ch * 49, Hd, K, T).to(gpu_device)
    x = torch.randn(R, D_ch, 7, 7, device=gpu_device) * 1.5
    monkeypatch.setenv("SNN_BF16X3_WN", tile[0])
    monkeypatch.setenv("SNN_BF16X3_MT", tile[1])

    def run():
        m.spike_rates = False
        c, b = m(x)
        m.spike_rates = True
        rates = m(x)
        return [c.clone(), b.clone()] + [r.clone() for r in rates] + [t.clone() for t in m.last_spike_counts]
    rm, wm = _both(monkeypatch, run)
    for a, b in zip(rm, wm):
        assert torch.equal(a, b)


def test_det_head_roialign_word_major_equals_row_major(gpu_device, monkeypatch):
    import snn_automotive_object_detection_amd as S
    from tests.test_gpu_roialign import _setup
    for R, C, T in ((200, 16, 12), (66, 32, 8), (2, 8, 5)):
        pool, feats, boxes, shapes = _setup(gpu_device, R=max(R, 6), C=C, seed=R)
        torch.manual_seed(R)
        head = S.FastRCNNPredictorSNNFull(C * 49, 128, 9, T).to(gpu_device)
        flist, scales, rois, lvl = pool.assign(feats, boxes, shapes)

        def run():
            c, b = head.forward_roialign(flist, scales, rois, lvl)
            return [c.clone(), b.clone()]
        rm, wm = _both(monkeypatch, run)
        assert torch.equal(rm[0], wm[0]) and torch.equal(rm[1], wm[1])


def test_misaligned_or_odd_width_rows_fall_back_to_row_major(gpu_device):
    """D % 32 != 0 (or rows not 16-byte aligned): the detector head keeps row-major planes and still matches the oracle path
    (the ordinary module tests cover the values; here: it runs and equals the aligned computation on the same data)"""
    import snn_automotive_object_detection_amd as S
    torch.manual_seed(2)
    m = S.FastRCNNPredictorSNNFull(3 * 49, 64, 5, 8).to(gpu_device)          # D = 147: not a multiple of 32
    x = torch.randn(50, 3, 7, 7, device=gpu_device)
    c, b = m(x)
    buf = torch.zeros(50 * 147 + 1, device=gpu_device)
    xs = buf[1:].view(50, 3, 7, 7)                                           # 4-byte aligned only
    xs.copy_(x)
    c2, b2 = m(xs)
    assert torch.equal(c, c2) and torch.equal(b, b2)


@pytest.mark.parametrize("T", [8, 12, 24])
@pytest.mark.parametrize("wn", ["1", "2"])
def test_rpn_spike_planes_in_blocks_of_four_words_equal_plain_rows(gpu_device, monkeypatch, T, wn):
    """conv -> LI heads hand-over at C = 256: planes [T][word / 4][position][4] (Gemm3Args.out_split, whole 32-byte sectors
    leave the L2) against plain rows [T][position][8] (SNN_SPK_SPLIT=0): same outputs, spike counts and rates, bit for bit"""
    import snn_automotive_object_detection_amd as S
    torch.manual_seed(T)
    m = S.RPNHeadSNN(256, 3, T).to(gpu_device)
    with torch.no_grad():
        m.shared_conv.weight.mul_(5.0)
    m.spike_rates = True
    feats = [torch.randn(n, 256, h, w, device=gpu_device) * 1.5 for n, h, w in [(2, 31, 47), (1, 9, 5), (2, 2, 3)]]
    monkeypatch.setenv("SNN_BF16X3_WN", wn)

    def run():
        lg, bb, rates = m(feats)
        return [x.clone() for x in lg + bb + list(rates)] + [m.last_spike_counts.clone()]
    monkeypatch.setenv("SNN_SPK_SPLIT", "0")
    plain = run()
    monkeypatch.delenv("SNN_SPK_SPLIT")
    split = run()
    assert int(plain[-1].sum()) > 0
    for a, b in zip(plain, split):
        assert torch.equal(a, b)
