"""The RPN's shared 3x3 convolution with its sparse period planes e_3 .. on the structured-sparse matrix-core instruction
(csrc/snn_sparse.h; default) against the all-dense launch (SNN_SPARSE=0) and the oracle (rpn.py:98-119):
 * same results up to the fp32 summation order: outputs within 1e-4 except at positions where a threshold tie flipped a spike
   (the usual budget), on pyramids of several T and odd shapes;
 * nibbles with three and four spikes of one period (more than one instruction can take) go through the secondary compressed
   plane: inputs built to have them in a few places, and nearly everywhere, in a sparse plane still match the dense launch - and
   repeat bit for bit;
 * (round 5) spike-rate mode really takes the structured-sparse COUNTING launches (rpn.py:126-200, faster_rcnn.py:520-618): the path
   is asserted, the integer counts are the popcounts of the spike planes the same launch wrote, equal the all-dense launch's counts
   wherever the two runs' planes agree, and the oracle's within the flip budget;
 * (round 5) fc6 beyond 16 steps (T_det = 17 .. 26, config[4]'s 24) and in spike-rate mode (window T - 1) runs the sparse launch with the
   general LIF epilogue."""
import numpy as np
import os

import pytest
import torch

from oracle import snn_oracle as OR
from tests._util import flip_budget

pytestmark = pytest.mark.gpu


def _head(dev, C, T, seed, gain=4.0):
    import snn_automotive_object_detection_amd as S
    torch.manual_seed(seed)
    m = S.RPNHeadSNN(C, 3, T).to(dev)
    with torch.no_grad():
        m.shared_conv.weight.mul_(gain)
    return m


def _run(m, feats, sparse=None):
    from snn_automotive_object_detection_amd import _lib
    lg, bb = m(feats)
    if sparse is not None:
        assert _lib.load().snn_debug_last_conv_path() == int(sparse)
    return [t.clone() for t in lg + bb]


def _off_positions(a, b, tol=1e-4):
    """positions (n, y, x) where any output channel differs by more than tol, and the largest difference"""
    bad, mx = 0, 0.0
    L = len(a) // 2
    for l in range(L):
        d = torch.maximum((a[l] - b[l]).abs().amax(1), (a[L + l] - b[L + l]).abs().amax(1))
        bad += int((d > tol).sum())
        mx = max(mx, float(d.max()))
    return bad, mx


@pytest.mark.parametrize("T", [5, 6, 7, 8, 9, 10, 12, 14, 16])
def test_sparse_conv_equals_dense_up_to_ties(gpu_device, monkeypatch, T):
    m = _head(gpu_device, 256, T, T)
    g = torch.Generator().manual_seed(T)
    feats = [(torch.randn(2, 256, h, w, generator=g) * 1.7).to(gpu_device) for h, w in [(37, 53), (19, 27), (7, 9), (1, 3)]]
    a = _run(m, feats, sparse=True)
    monkeypatch.setenv("SNN_SPARSE", "0")
    b = _run(m, feats, sparse=False)
    pos = sum(2 * f.shape[2] * f.shape[3] for f in feats)
    bad, mx = _off_positions(a, b)
    assert bad <= flip_budget(pos, 256, T, "rpn_randn") and mx < 0.05, (bad, mx)
    assert any(float(t.abs().max()) > 0 for t in a)


def test_sparse_conv_vs_oracle(gpu_device):
    T = 8
    m = _head(gpu_device, 64, T, 3)
    g = torch.Generator().manual_seed(5)
    feats = [torch.randn(2, 64, 23, 31, generator=g) * 1.7, torch.randn(2, 64, 6, 5, generator=g) * 1.7]
    a = _run(m, [f.to(gpu_device) for f in feats], sparse=True)
    o_l, o_b = OR.rpn_head_forward(feats, m.shared_conv.weight.detach().cpu(), m.conv_cls.weight.detach().cpu(), m.conv_bbox.weight.detach().cpu(), T)
    bad, mx = _off_positions([t.cpu() for t in a], list(o_l) + list(o_b))
    assert bad <= flip_budget(2 * (23 * 31 + 30), 64, T, "rpn_randn") and mx < 0.05, (bad, mx)


def _same_period_blocks(C, H, W, period, frac, seed):
    """features whose channels fire with one period in whole groups of four adjacent channels (3 - 4 spikes per nibble of plane e_period)
    on a fraction of the positions; the rest N(0, 1.7)"""
    import ctypes as Ct
    from snn_automotive_object_detection_amd import _lib, ops
    p = ops.make_params(ops.LIFParameters(v_th=torch.tensor(0.25)), ops.LIFParameters(alpha=100, v_th=torch.tensor(0.1)))
    th = (Ct.c_float * 32)()
    assert _lib.load().snn_debug_encoder_thresholds(Ct.byref(p), th) == 1
    x_n = 0.5 * (th[period - 1] + th[period - 2])            # first spike at step period - 1  <=>  th[period - 1] <= x < th[period - 2]
    g = torch.Generator().manual_seed(seed)
    f = torch.randn(2, C, H, W, generator=g) * 1.7
    mask = torch.rand(2, 1, H, W, generator=g) < frac
    grp = (torch.rand(2, C // 4, H, W, generator=g) < 0.5).repeat_interleave(4, dim=1)     # whole nibbles
    three = torch.rand(2, C, H, W, generator=g) < 0.9                                      # mostly 4, sometimes 3 of a nibble
    sel = mask & grp & three
    return torch.where(sel, torch.full_like(f, float(x_n)), f)


@pytest.mark.parametrize("period,frac", [(3, 0.02), (5, 0.05), (7, 0.03), (4, 0.9), (3, 1.0)])
def test_secondary_plane_carries_the_spikes_one_instruction_cannot(gpu_device, monkeypatch, period, frac):
    """(frac 0.9 / 1.0: nearly every 16-row M-tile of plane e_period needs its second instruction in every step)"""
    T = 8
    m = _head(gpu_device, 128, T, period)
    feats = [_same_period_blocks(128, 20, 30, period, frac, period).to(gpu_device), _same_period_blocks(128, 5, 7, period, frac, period + 1).to(gpu_device)]
    a = _run(m, feats, sparse=True)
    monkeypatch.setenv("SNN_SPARSE", "0")
    b = _run(m, feats, sparse=False)
    bad, mx = _off_positions(a, b)
    assert bad <= 2 + flip_budget(2 * 635, 128, T, "rpn_randn") and mx < 0.05, (bad, mx)
    assert any(float(t.abs().max()) > 0 for t in a)
    monkeypatch.delenv("SNN_SPARSE")
    for _ in range(3):
        assert all(torch.equal(x, y) for x, y in zip(a, _run(m, feats)))


# ---- the detector head's fc6 (faster_rcnn.py:498-499) on the same instruction ---------------------------------------------------
def _det(dev, C, Hd, K, T, seed):
    import snn_automotive_object_detection_amd as S
    torch.manual_seed(seed)
    d = S.FastRCNNPredictorSNNFull(C * 49, Hd, K, T).to(dev)
    with torch.no_grad():
        d.fc7.weight.mul_(3.0)
    return d


def _run_det(d, x, sparse=None):
    from snn_automotive_object_detection_amd import _lib
    c, b = d(x)
    if sparse is not None:
        assert _lib.load().snn_debug_last_fc6_path() == int(sparse)
    return c.clone(), b.clone()


@pytest.mark.parametrize("R,C,Hd,K,T", [(2000, 256, 1024, 9, 12), (333, 64, 256, 11, 16), (45, 64, 128, 5, 8), (1, 64, 128, 5, 6), (130, 128, 192, 3, 12)])
def test_sparse_fc6_equals_dense_up_to_ties(gpu_device, monkeypatch, R, C, Hd, K, T):
    d = _det(gpu_device, C, Hd, K, T, R)
    x = torch.randn(R, C, 7, 7, device=gpu_device) * 2
    a = _run_det(d, x, sparse=True)
    monkeypatch.setenv("SNN_SPARSE", "0")
    b = _run_det(d, x, sparse=False)
    off = ((a[0] - b[0]).abs().amax(1) > 1e-4) | ((a[1] - b[1]).abs().amax(1) > 1e-4)
    assert int(off.sum()) <= flip_budget(R, 2 * Hd, T, "det") and float((a[0] - b[0]).abs().max()) < 0.05, int(off.sum())
    assert float(a[0].abs().max()) > 0


def test_fc6_reduction_order_permutation_is_the_same_contraction(gpu_device, monkeypatch):
    """fc6's weights in the order k' = bin * C + channel, the encoder's planes transposed to match (k_permute_planes): against the
    reference's order (SNN_FC6_PERM=0) the same sums in another order - on both the dense launch and the fused RoIAlign path"""
    from tests.test_gpu_roialign import _setup
    monkeypatch.setenv("SNN_SPARSE", "0")
    d = _det(gpu_device, 64, 256, 9, 12, 4)
    assert d.fc6_inner() == 49
    x = torch.randn(300, 64, 7, 7, device=gpu_device) * 2
    pool, fm, boxes, shapes = _setup(gpu_device, R=200, C=64, seed=4)
    flist, scales, rois, lvl = pool.assign(fm, boxes, shapes)
    a = _run_det(d, x) + tuple(t.clone() for t in d.forward_roialign(flist, scales, rois, lvl))
    monkeypatch.setenv("SNN_FC6_PERM", "0")
    d.invalidate_packed_weights()
    assert d.fc6_inner() == 0
    b = _run_det(d, x) + tuple(t.clone() for t in d.forward_roialign(flist, scales, rois, lvl))
    for i in (0, 2):
        off = ((a[i] - b[i]).abs().amax(1) > 1e-4) | ((a[i + 1] - b[i + 1]).abs().amax(1) > 1e-4)
        assert int(off.sum()) <= flip_budget(a[i].shape[0], 2 * 256, 12, "det") and float((a[i] - b[i]).abs().max()) < 0.05
    # a head whose channel count is not a multiple of 32 keeps the reference's order
    assert _det(gpu_device, 8, 64, 5, 6, 1).fc6_inner() == 0


def test_sparse_fc6_vs_oracle(gpu_device):
    T, C, Hd, K, R = 12, 64, 128, 9, 60
    d = _det(gpu_device, C, Hd, K, T, 2)
    g = torch.Generator().manual_seed(8)
    x = torch.randn(R, C, 7, 7, generator=g) * 2
    a = _run_det(d, x.to(gpu_device), sparse=True)
    o_c, o_d = OR.det_head_forward(x, d.fc6.weight.detach().cpu(), d.fc7.weight.detach().cpu(), d.cls_score.weight.detach().cpu(),
                                   d.bbox_pred.weight.detach().cpu(), T)
    off = ((a[0].cpu() - o_c).abs().amax(1) > 1e-4) | ((a[1].cpu() - o_d).abs().amax(1) > 1e-4)
    assert int(off.sum()) <= flip_budget(R, 2 * Hd, T, "det"), int(off.sum())


@pytest.mark.parametrize("frac", [0.003, 0.6, 1.0])
def test_sparse_fc6_secondary_plane(gpu_device, monkeypatch, frac):
    """whole nibbles of features firing with one period (3 - 4 spikes per nibble of a sparse plane), a few and nearly all: the
    secondary compressed plane carries them"""
    T, C, Hd, K, R = 12, 64, 128, 9, 100
    d = _det(gpu_device, C, Hd, K, T, 5)
    f = _same_period_blocks(C, 7, 7 * R // 2, 5, frac, 3)                       # [2, C, 7, 7 R / 2] -> R RoIs of [C, 7, 7]
    x = f.reshape(2, C, 7, R // 2, 7).permute(0, 3, 1, 2, 4).reshape(R, C, 7, 7).contiguous().to(gpu_device)
    a = _run_det(d, x, sparse=True)
    again = _run_det(d, x, sparse=True)
    assert torch.equal(a[0], again[0]) and torch.equal(a[1], again[1])
    monkeypatch.setenv("SNN_SPARSE", "0")
    b = _run_det(d, x, sparse=False)
    off = ((a[0] - b[0]).abs().amax(1) > 1e-4) | ((a[1] - b[1]).abs().amax(1) > 1e-4)
    assert int(off.sum()) <= flip_budget(R, 2 * Hd, T, "det") and float((a[0] - b[0]).abs().max()) < 0.05
    assert float(a[0].abs().max()) > 0


# ---- spike-rate mode on the sparse launches (VERDICT r4 P-a) ------------------------------------------------------------------
def _popcount(t):
    lut = torch.tensor([bin(i).count("1") for i in range(256)], dtype=torch.int64, device=t.device)
    return lut[t.contiguous().view(torch.uint8).to(torch.int64)]


def _rpn_hidden_planes(dev, T, Cw):
    """the shared LIF's spike planes [T, P, Cw] the calling thread's last RPN head left in its workspace"""
    import ctypes as Ct
    from snn_automotive_object_detection_amd import _lib, ops
    off3 = (Ct.c_uint64 * 3)()
    _lib.load().snn_debug_last_rpn_planes(off3)
    P = int(off3[2])
    raw = ops._WS.get(dev, 1)[int(off3[0]): int(off3[0]) + T * P * Cw * 4].view(torch.int32)
    planes = raw.view(T, Cw // 4, P, 4).permute(0, 2, 1, 3).reshape(T, P, Cw) if off3[1] else raw.view(T, P, Cw)
    return planes.clone()


@pytest.mark.parametrize("T", [8, 16])
def test_spike_rate_mode_runs_the_sparse_counting_conv(gpu_device, monkeypatch, T):
    from snn_automotive_object_detection_amd import _lib
    C, shapes = 256, [(37, 53), (19, 27), (7, 9)]
    m = _head(gpu_device, C, T, 40 + T)
    g = torch.Generator().manual_seed(40 + T)
    feats = [torch.randn(2, C, h, w, generator=g) * 1.7 for h, w in shapes]
    gold = []
    OR.rpn_head_forward(feats, m.shared_conv.weight.detach().cpu(), m.conv_cls.weight.detach().cpu(), m.conv_bbox.weight.detach().cpu(), T,
                        spike_rates=True, counts_out=gold)
    fd = [f.to(gpu_device) for f in feats]
    m.spike_rates = True
    runs = {}
    for sparse in (True, False):
        if not sparse:
            monkeypatch.setenv("SNN_SPARSE", "0")
        m(fd)
        assert _lib.load().snn_debug_last_conv_path() == int(sparse)          # counting launches stay on the sparse kernel
        runs[sparse] = (m.last_spike_counts.clone().cpu(), _rpn_hidden_planes(gpu_device, T, C // 32))
    base, n_differ = 0, 0
    for l, (h, w) in enumerate(shapes):
        for n in range(2):
            sl = slice(base + n * h * w, base + (n + 1) * h * w)
            pops = {k: int(_popcount(runs[k][1][:, sl]).sum()) for k in runs}
            for k in runs:                                                     # a launch's count IS the popcount of the planes it wrote
                assert int(runs[k][0][l, n]) == pops[k], (k, l, n, int(runs[k][0][l, n]), pops[k])
            differ = int((runs[True][1][:, sl] != runs[False][1][:, sl]).any(0).any(1).sum())      # positions whose trains differ (tie flips)
            n_differ += differ
            if differ == 0:
                assert int(runs[True][0][l, n]) == int(runs[False][0][l, n])
            assert abs(int(runs[True][0][l, n]) - int(gold[l][n])) <= 4 * flip_budget(h * w, C, T, "rpn_randn"), (l, n)
        base += 2 * h * w
    assert n_differ <= flip_budget(base, C, T, "rpn_randn"), n_differ
    assert int(runs[True][0].sum()) > 0


@pytest.mark.parametrize("T", [8, 12, 16, 24])
def test_spike_rate_mode_runs_the_sparse_counting_fc6(gpu_device, monkeypatch, T):
    import ctypes as Ct
    from snn_automotive_object_detection_amd import _lib, ops
    R, C, Hd, K = 333, 64, 256, 9
    d = _det(gpu_device, C, Hd, K, T, 50 + T)
    g = torch.Generator().manual_seed(50 + T)
    x = torch.randn(R, C, 7, 7, generator=g) * 2
    gold = []
    OR.det_head_forward(x, d.fc6.weight.detach().cpu(), d.fc7.weight.detach().cpu(), d.cls_score.weight.detach().cpu(), d.bbox_pred.weight.detach().cpu(), T,
                        spike_rates=True, counts_out=gold)
    xd = x.to(gpu_device)
    d.spike_rates = True
    runs = {}
    for sparse in (True, False):
        if not sparse:
            monkeypatch.setenv("SNN_SPARSE", "0")
        d(xd)
        assert _lib.load().snn_debug_last_fc6_path() == int(sparse)
        off3 = (Ct.c_uint64 * 3)()
        _lib.load().snn_debug_last_det_planes(off3)
        assert off3[2] == 1                                                    # lif6's planes word-major [T][Hd / 32][R]
        s6 = ops._WS.get(gpu_device, 1)[int(off3[0]): int(off3[0]) + T * (Hd // 32) * R * 4].view(torch.int32).view(T, Hd // 32, R).clone()
        runs[sparse] = (d.last_spike_counts[0].clone(), d.last_spike_counts[1].clone(), s6)
    for k in runs:                                                             # per-RoI count == popcount of the RoI's lif6 planes of the same launch
        assert torch.equal(runs[k][0].to(torch.int64), _popcount(runs[k][2]).view(T, Hd // 32, R, 4).sum(dim=(0, 1, 3))), k
    same = (runs[True][2] == runs[False][2]).all(0).all(0)                     # RoIs whose lif6 trains agree between the two launches
    assert torch.equal(runs[True][0][same], runs[False][0][same])
    assert int((~same).sum()) <= flip_budget(R, Hd, T, "det"), int((~same).sum())
    off_gold = int((runs[True][0].cpu() != gold[0]).sum())
    assert off_gold <= flip_budget(R, Hd, T, "det"), off_gold
    assert int(runs[True][0].sum()) > 0


@pytest.mark.parametrize("T", [17, 20, 24, 26])
def test_sparse_fc6_beyond_16_steps(gpu_device, monkeypatch, T):
    """T_det = 17 .. 26 (train.py:40-43 default 16, config[4] 24): 15 .. 24 period planes of 16 RoIs on the 4 x 2 wave grid, general LIF epilogue"""
    R, C, Hd, K = 150, 64, 256, 9
    d = _det(gpu_device, C, Hd, K, T, 60 + T)
    g = torch.Generator().manual_seed(60 + T)
    x = torch.randn(R, C, 7, 7, generator=g) * 2
    a = _run_det(d, x.to(gpu_device), sparse=True)
    o_c, o_d = OR.det_head_forward(x, d.fc6.weight.detach().cpu(), d.fc7.weight.detach().cpu(), d.cls_score.weight.detach().cpu(),
                                   d.bbox_pred.weight.detach().cpu(), T)
    off = ((a[0].cpu() - o_c).abs().amax(1) > 1e-4) | ((a[1].cpu() - o_d).abs().amax(1) > 1e-4)
    assert int(off.sum()) <= flip_budget(R, 2 * Hd, T, "det"), int(off.sum())
    assert float(a[0].abs().max()) > 0
    again = _run_det(d, x.to(gpu_device), sparse=True)
    assert torch.equal(a[0], again[0]) and torch.equal(a[1], again[1])
    monkeypatch.setenv("SNN_SPARSE", "0")
    b = _run_det(d, x.to(gpu_device), sparse=False)
    off = ((a[0] - b[0]).abs().amax(1) > 1e-4) | ((a[1] - b[1]).abs().amax(1) > 1e-4)
    # (a RoI whose tie fell the other way drifts further over 26 steps than over 12: 0.07 observed at T = 26 with fc7's weights x 3)
    assert int(off.sum()) <= flip_budget(R, 2 * Hd, T, "det") and float((a[0] - b[0]).abs().max()) < 0.5, int(off.sum())


# ---- the FAT shape of linear layers (work-groups of four waves with twice the M-tile slots per wave; default): the same sums in the same
# order as the 8-wave shape (SNN_SPARSE_FAT=0) - every accumulator sees the same matrix instructions in the same order -------------
# (fat = 0: a plan whose row-waves have no FAT loop instance - four dense M-tiles per row-wave at T <= 8 - keeps the 8-wave shape)
@pytest.mark.parametrize("R,C,Hd,K,T,fat", [(2000, 256, 1024, 9, 12, 1), (333, 64, 256, 11, 24, 1), (77, 64, 128, 5, 12, 1), (1, 64, 128, 5, 24, 1), (200, 128, 192, 3, 16, 1),
                                            (90, 64, 128, 5, 7, 0), (2000, 256, 1024, 9, 8, 0), (500, 64, 256, 9, 19, 1), (2000, 256, 1024, 9, 14, 1)])
def test_fat_shape_fc6_is_bit_identical(gpu_device, monkeypatch, R, C, Hd, K, T, fat):
    d = _det(gpu_device, C, Hd, K, T, R + T)
    x = torch.randn(R, C, 7, 7, device=gpu_device) * 2
    monkeypatch.setenv("SNN_SPARSE_FAT", "0")
    a = _run_det(d, x, sparse=True)
    monkeypatch.delenv("SNN_SPARSE_FAT")
    import ctypes as Ct
    from snn_automotive_object_detection_amd import _lib
    o12 = (Ct.c_int32 * 12)()
    assert _lib.load().snn_debug_tile_shape(0, R, C * 49, Hd, T, 0, 6, o12) == 0 and o12[8] == 1 and o12[1] == fat, list(o12)     # the FAT plan is the default
    for _ in range(3):
        b = _run_det(d, x, sparse=True)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert float(a[0].abs().max()) > 0 or T < 12
    d.spike_rates = True
    d(x)
    c_big = [c.clone() for c in d.last_spike_counts]
    monkeypatch.setenv("SNN_SPARSE_FAT", "0")
    d(x)
    assert all(torch.equal(p, q) for p, q in zip(c_big, d.last_spike_counts))


# ---- plane compression inside the RPN encoder launch (round 5): the same compressed planes as the separate k_compress_planes pass ----------
@pytest.mark.parametrize("C,T,shapes", [(256, 8, [(37, 53), (19, 27), (7, 9), (1, 3)]), (64, 5, [(23, 31), (6, 5)]), (128, 16, [(20, 30), (5, 7), (1, 1)]),
                                        (256, 12, [(64, 64)]), (512, 8, [(9, 11), (3, 5)])])
def test_encoder_fold_writes_the_same_compressed_planes(gpu_device, monkeypatch, C, T, shapes):
    from snn_automotive_object_detection_amd import _lib, ops
    m = _head(gpu_device, C, T, 80 + T)
    g = torch.Generator().manual_seed(80 + T)
    feats = [(torch.randn(2, C, h, w, generator=g) * 1.7).to(gpu_device) for h, w in shapes]
    al = lambda x: (x + 255) // 256 * 256
    Cw, Pe = C // 32, sum(2 * (h + 2) * (w + 2) for h, w in shapes)
    o_cur = 2 * al(T * Pe * Cw * 4)                             # rpn_ws_layout: encoder planes, spike planes, then the sparse conv's side buffers
    cmp_bytes = (T - 1 - 2) * (Cw // 2) * 4 * Pe * 4

    def run():
        ops._WS.get(gpu_device, 1)[o_cur: o_cur + cmp_bytes].fill_(0x5a) if ops._WS.get(gpu_device, 1).numel() >= o_cur + cmp_bytes else None
        out = _run(m, feats, sparse=True)
        return out, ops._WS.get(gpu_device, 1)[o_cur: o_cur + cmp_bytes].clone()
    run()                                                       # (sizes the workspace)
    monkeypatch.setenv("SNN_ENC_FOLD", "0")
    a, cmp_a = run()
    monkeypatch.delenv("SNN_ENC_FOLD")
    for _ in range(2):
        b, cmp_b = run()
        assert torch.equal(cmp_a, cmp_b), int((cmp_a != cmp_b).sum())
        assert all(torch.equal(x, y) for x, y in zip(a, b))
    assert int((cmp_a.view(torch.int32)[: (Cw // 2) * 4 * Pe].view(-1, Pe)[0] != 0).sum()) > 0      # occupancy words of plane e_3: something fired
    m.spike_rates = True                                        # the counting launches read the same planes
    m(feats)
    c_fold = m.last_spike_counts.clone()
    monkeypatch.setenv("SNN_ENC_FOLD", "0")
    m(feats)
    assert torch.equal(c_fold, m.last_spike_counts)


@pytest.mark.parametrize("R,C,Hd,K,T", [(2000, 256, 1024, 9, 12), (333, 64, 256, 11, 24), (77, 64, 128, 5, 12), (1, 64, 128, 5, 8), (50, 128, 192, 3, 16), (19, 192, 64, 3, 26)])
def test_detector_encoder_fold_writes_the_same_planes(gpu_device, monkeypatch, R, C, Hd, K, T):
    """k_encode_rows_perm (encoder + fc6's reduction order + compression in one launch) against k_encode_rows_wm -> k_permute_planes ->
    k_compress_planes (SNN_ENC_FOLD=0): the dense planes e_1, e_2 and the compressed planes e_3 .. in the workspace bit for bit, and the
    head's outputs and spike counts with them"""
    from snn_automotive_object_detection_amd import ops
    d = _det(gpu_device, C, Hd, K, T, R + 3 * T)
    x = torch.randn(R, C, 7, 7, device=gpu_device) * 2
    al = lambda v: (v + 255) // 256 * 256
    Dw, Tc = C * 49 // 32, T - 2
    o_cur = al(T * R * Dw * 4)                                  # det_ws_layout: the encoder planes (fc6 reads them at the front), then the side buffers
    dense_bytes, cmp_bytes = 2 * Dw * R * 4, (Tc - 2) * (Dw // 2) * 4 * R * 4

    def run():
        out = _run_det(d, x, sparse=True)
        ws = ops._WS.get(gpu_device, 1)
        return out, ws[:dense_bytes].clone(), ws[o_cur: o_cur + cmp_bytes].clone()
    run()
    monkeypatch.setenv("SNN_ENC_FOLD", "0")
    a, dense_a, cmp_a = run()
    monkeypatch.delenv("SNN_ENC_FOLD")
    ops._WS.get(gpu_device, 1)[: o_cur + cmp_bytes].fill_(0x5a)
    for _ in range(2):
        b, dense_b, cmp_b = run()
        assert torch.equal(dense_a, dense_b), int((dense_a != dense_b).sum())
        assert torch.equal(cmp_a, cmp_b), int((cmp_a != cmp_b).sum())
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert int((dense_a != 0).sum()) > 0
    d.spike_rates = True                                        # (window T - 1: one more compressed plane)
    d(x)
    c_fold = [c.clone() for c in d.last_spike_counts]
    monkeypatch.setenv("SNN_ENC_FOLD", "0")
    d(x)
    assert all(torch.equal(p, q) for p, q in zip(c_fold, d.last_spike_counts))


@pytest.mark.parametrize("R,C,T", [(333, 64, 12), (77, 128, 16), (40, 64, 26), (2000, 256, 12)])
def test_detector_encoder_block_shapes_write_the_same_planes(gpu_device, monkeypatch, R, C, T):
    """k_encode_rows_perm on 8 / 16 RoIs x 4 / 8 waves per block (SNN_ENCP_RB, SNN_ENCP_NW; 16 RoIs with more than 12 planes: two passes
    through LDS, the second re-deriving its cumulative words): the dense and the compressed planes in the workspace bit for bit"""
    from snn_automotive_object_detection_amd import ops
    d = _det(gpu_device, C, 128, 5, T, R + T)
    x = torch.randn(R, C, 7, 7, device=gpu_device) * 2
    al = lambda v: (v + 255) // 256 * 256
    Dw, Tc = C * 49 // 32, T - 2
    o_cur = al(T * R * Dw * 4)
    dense_bytes, cmp_bytes = 2 * Dw * R * 4, (Tc - 2) * (Dw // 2) * 4 * R * 4

    def run():
        out = _run_det(d, x, sparse=True)
        ws = ops._WS.get(gpu_device, 1)
        return out, ws[:dense_bytes].clone(), ws[o_cur: o_cur + cmp_bytes].clone()
    ref = run()
    for rb in (8, 16):
        for nw in (4, 8):
            monkeypatch.setenv("SNN_ENCP_RB", str(rb))
            monkeypatch.setenv("SNN_ENCP_NW", str(nw))
            ops._WS.get(gpu_device, 1)[: o_cur + cmp_bytes].fill_(0x5a)
            got = run()
            assert torch.equal(ref[1], got[1]) and torch.equal(ref[2], got[2]), (rb, nw)
            assert torch.equal(ref[0][0], got[0][0]) and torch.equal(ref[0][1], got[0][1]), (rb, nw)


# ---- the FAT conv (four waves, each ALL planes of its own 16 positions, the LIF in registers - no tile image) ---------------------------
@pytest.mark.parametrize("T", [7, 8, 9, 10, 12, 13, 16])
@pytest.mark.parametrize("C", [256, 64, 128])
def test_fat_conv_register_lif_is_bit_identical(gpu_device, monkeypatch, T, C):
    """same matrix instructions per accumulator, same LIF operations in the same order: the spike planes - and so the outputs and the
    integer spike counts of spike-rate mode - equal the 8-wave shape's bit for bit (pyramid with partial tiles and tiles that straddle
    levels).  T <= 9: 4 x 1 waves, tiles of 64 positions; T = 12 .. 16: 2 x 2 waves, tiles of 32"""
    import ctypes as Ct
    from snn_automotive_object_detection_amd import _lib
    m = _head(gpu_device, C, T, 90 + T)
    g = torch.Generator().manual_seed(90 + T + C)
    feats = [(torch.randn(2, C, h, w, generator=g) * 1.7).to(gpu_device) for h, w in [(41, 67), (19, 27), (7, 9), (1, 3)]]
    monkeypatch.setenv("SNN_SPARSE_FAT_CONV", "0")
    a = _run(m, feats, sparse=True)
    planes_a = _rpn_hidden_planes(gpu_device, T, C // 32)
    monkeypatch.setenv("SNN_SPARSE_FAT_CONV", "3")
    o12 = (Ct.c_int32 * 12)()
    pos = sum(2 * f.shape[2] * f.shape[3] for f in feats)
    fat = T <= 9 or T >= 12                                     # (T = 10, 11: the 8-wave shape's tiles of 48 positions are faster)
    for rates in (0, 1):
        assert _lib.load().snn_debug_tile_shape(1, pos, C, C, T, rates, 0, o12) == 0 and o12[8] == 1 and o12[1] == int(fat), list(o12)
        assert o12[3] == (64 if T <= 9 else 32 if fat else 48), list(o12)
    for _ in range(3):
        b = _run(m, feats, sparse=True)
        planes_b = _rpn_hidden_planes(gpu_device, T, C // 32)
        assert torch.equal(planes_a, planes_b), int((planes_a != planes_b).sum())
        assert all(torch.equal(x, y) for x, y in zip(a, b))
    assert int((planes_a != 0).sum()) > 0
    m.spike_rates = True                                        # the register LIF counts spikes too (per-position atomics, then k_sum_pos_counts)
    m(feats)
    c1 = m.last_spike_counts.clone()
    monkeypatch.setenv("SNN_SPARSE_FAT_CONV", "0")
    m(feats)
    assert torch.equal(c1, m.last_spike_counts) and int(c1.sum()) > 0


# ---- the ping-pong form of the FAT conv (round 6: one persistent work-group of 8 waves per CU, the two waves of a SIMD alternate) ----------
@pytest.mark.skipif("PP" not in os.path.basename(os.environ.get("SNN_HIP_LIB", "")), reason="needs a -DSNN_PINGPONG build (bash tools/ab_build.sh PP:\"-DSNN_PINGPONG\"; "
                    "SNN_HIP_LIB=tools/_ab/lib_PP.so): the ping-pong conv measured slower than the FAT conv and is not in the product library")
@pytest.mark.parametrize("T", [7, 8])
@pytest.mark.parametrize("C,shapes", [(256, [(41, 67), (19, 27), (7, 9), (1, 3)]), (64, [(41, 67), (19, 27), (7, 9), (1, 3)]), (128, [(5, 7)]),
                                      (256, [(192, 384), (96, 192), (48, 96), (24, 48), (12, 24)])])
def test_ping_pong_conv_is_bit_identical(gpu_device, monkeypatch, T, C, shapes):
    """k_conv_lif_pp (csrc/snn_sparse_pp.h) against k_gemm_lif_sparse<true, 1, FAT>: same matrix instructions per accumulator, same register LIF -
    the hidden spike planes, the outputs and the integer spike counts of spike-rate mode bit for bit; pyramids with partial tiles, tiles that
    straddle levels, fewer tile pairs than work-groups, an odd number of tiles per XCD, and the full Cityscapes pyramid"""
    m = _head(gpu_device, C, T, 190 + T)
    g = torch.Generator().manual_seed(190 + T + C)
    feats = [(torch.randn(2, C, h, w, generator=g) * 1.7).to(gpu_device) for h, w in shapes]
    monkeypatch.setenv("SNN_CONV_PP", "0")
    a = _run(m, feats, sparse=True)
    planes_a = _rpn_hidden_planes(gpu_device, T, C // 32)
    monkeypatch.setenv("SNN_CONV_PP", "1")
    for _ in range(3):
        b = _run(m, feats, sparse=True)
        planes_b = _rpn_hidden_planes(gpu_device, T, C // 32)
        assert torch.equal(planes_a, planes_b), int((planes_a != planes_b).sum())
        assert all(torch.equal(x, y) for x, y in zip(a, b))
    assert int((planes_a != 0).sum()) > 0
    m.spike_rates = True
    m(feats)
    c1 = m.last_spike_counts.clone()
    monkeypatch.setenv("SNN_CONV_PP", "0")
    m(feats)
    assert torch.equal(c1, m.last_spike_counts) and int(c1.sum()) > 0


@pytest.mark.parametrize("R,C,Hd,K,T", [(2000, 256, 1024, 9, 12), (2000, 256, 1024, 9, 14), (333, 64, 256, 11, 11), (77, 64, 128, 5, 12), (31, 64, 128, 5, 9), (1, 64, 128, 5, 13)])
def test_fat_fc6_register_lif_is_bit_identical(gpu_device, monkeypatch, R, C, Hd, K, T):
    """fc6 on the FAT shape with the LIF in registers (each row-wave all planes of its own 16 RoIs) against the same shape through the LDS
    tile image (SNN_LIF_REGS=0): lif6's spike planes in the workspace and the head's outputs bit for bit"""
    import ctypes as Ct
    from snn_automotive_object_detection_amd import _lib, ops
    d = _det(gpu_device, C, Hd, K, T, R + 7 * T)
    x = torch.randn(R, C, 7, 7, device=gpu_device) * 2

    def run():
        out = _run_det(d, x, sparse=True)
        off3 = (Ct.c_uint64 * 3)()
        _lib.load().snn_debug_last_det_planes(off3)
        s6 = ops._WS.get(gpu_device, 1)[int(off3[0]): int(off3[0]) + T * (Hd // 32) * R * 4].clone()
        return out, s6
    monkeypatch.setenv("SNN_LIF_REGS", "0")
    a, s6_a = run()
    monkeypatch.delenv("SNN_LIF_REGS")
    for _ in range(3):
        b, s6_b = run()
        assert torch.equal(s6_a, s6_b), int((s6_a != s6_b).sum())
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert int((s6_a != 0).sum()) > 0
