"""BASELINE.json full sizes (Cityscapes 1024x2048 -> 768x1536 pyramid, b=2, T_rpn=8; 2000 RoIs, T_det=12):
free-running parity against the oracle on the host plus size-independent properties."""
import numpy as np
import pytest
import torch

from tests._util import TIE_MARGIN, first_flip_margins, flip_budget, nchw_to_rows, planes_to_dense, record_parity

pytestmark = pytest.mark.gpu
LEVELS = [(192, 384), (96, 192), (48, 96), (24, 48), (12, 24)]


MXFP6_SWEEP = pytest.param("mxfp6", marks=pytest.mark.sweep)     # the opt-in precision (not the headline) at full size: `-m "gpu and sweep"`


@pytest.mark.parametrize("precision", ["bf16x3", "f32", MXFP6_SWEEP])
def test_rpn_head_full_size_vs_oracle(gpu_device, monkeypatch, precision):
    import snn_automotive_object_detection_amd as S
    from oracle import snn_oracle as OR
    g = torch.Generator().manual_seed(7)
    feats = [torch.randn((2, 256, h, w), generator=g) for h, w in LEVELS]
    torch.manual_seed(1234)                                   # (the weights too: the result must not depend on which tests ran before)
    m = S.RPNHeadSNN(256, 3, 8)
    gold_counts = []
    with torch.no_grad():
        o_l, o_b = OR.rpn_head_forward(feats, m.shared_conv.weight, m.conv_cls.weight, m.conv_bbox.weight, 8, counts_out=gold_counts)
    m = m.to(gpu_device)
    m.spike_rates = True
    m.precision = precision
    logits, bbox, rates = m([f.to(gpu_device) for f in feats])
    head_l1 = None
    if precision == "bf16x3":
        # the spike planes THIS call left in the workspace (level 1): the head runs the structured-sparse counting launch, the stage-level
        # entry point used for the attribution below the all-dense one - another summation order, whose tie flips need not be the head's
        import ctypes as Ct
        from snn_automotive_object_detection_amd import _lib, ops
        off3 = (Ct.c_uint64 * 3)()
        _lib.load().snn_debug_last_rpn_planes(off3)
        P_all = int(off3[2])
        ws = ops._WS.get(gpu_device, 1)
        raw = ws[int(off3[0]): int(off3[0]) + 8 * P_all * 8 * 4].view(torch.int32)
        planes = raw.view(8, 2, P_all, 4).permute(0, 2, 1, 3).reshape(8, P_all, 8) if off3[1] else raw.view(8, P_all, 8)
        b1 = 2 * LEVELS[0][0] * LEVELS[0][1]
        head_l1 = planes[:, b1: b1 + 2 * LEVELS[1][0] * LEVELS[1][1]].contiguous().clone()
    total, bad = 0, 0
    for l in range(5):
        d = torch.maximum((logits[l].cpu() - o_l[l]).abs().amax(dim=1), (bbox[l].cpu() - o_b[l]).abs().amax(dim=1))
        total += d.numel()
        bad += int((d > 1e-4).sum())
        assert float(d.max()) < 0.05
    # threshold ties flip ~1e-7 of the spikes between two fp32 summation orders (SURVEY §7 risk 1)
    budget = flip_budget(total, 256, 8, "rpn_randn", precision)
    record_parity("rpn_head_full_size", precision=precision, positions_off_tolerance=bad, positions=total, budget=budget)
    assert bad <= budget, "positions off-tolerance: %d of %d" % (bad, total)
    # attribution on one level (96 x 192, b=2): every position off tolerance holds a flipped hidden spike, and every FIRST
    # flipped spike sits within TIE_MARGIN of the threshold in the oracle's trace (stage-level launch of the same kernels)
    if precision in ("bf16x3", "f32"):
        from snn_automotive_object_detection_amd import ops
        lvl = 1
        h, w = LEVELS[lvl]
        with torch.no_grad():
            _, _, tr = OR.rpn_head_forward([feats[lvl]], m.shared_conv.weight.cpu(), m.conv_cls.weight.cpu(), m.conv_bbox.weight.cpu(), 8, trace=True)
            _, _, vdec = OR.lif_scan_from_currents(tr[0]["cur"])
        p = m._params()
        if precision == "bf16x3":
            monkeypatch.setenv("SNN_STAGE_PERIODS", "1")       # what the bf16x3 head itself runs: period planes (csrc/snn_common.h)
        enc = ops.encode_nchw(feats[lvl].to(gpu_device), 8, p)
        if precision == "bf16x3":
            spk = ops.conv3x3_lif_bf16x3(enc, [(2, h, w)], 256, 256, p, m._packed_shared())
        else:
            spk = ops.conv3x3_lif(enc, 2, 256, 256, h, w, p, m._packed_shared())
        gold_spk, gold_vdec = nchw_to_rows(tr[0]["spk"]), nchw_to_rows(vdec)      # (kept for the second attribution below: the oracle's trace of this level costs ~8 s)
        n_flip, margins, flipped = first_flip_margins(planes_to_dense(spk, 256), gold_spk, gold_vdec)
        del tr, vdec
        d_l = torch.maximum((logits[lvl].cpu() - o_l[lvl]).abs().amax(dim=1), (bbox[lvl].cpu() - o_b[lvl]).abs().amax(dim=1)).reshape(-1).numpy()
        off_pos = np.nonzero(d_l > 1e-4)[0]
        flipped_pos = flipped.any(axis=1)                                   # [positions]
        record_parity("rpn_head_full_size_flips", precision=precision, level=lvl, flipped_neurons=n_flip,
                      worst_margin=float(margins.max()) if n_flip else 0.0, positions_off_tolerance=int(off_pos.size),
                      positions_with_flip=int(flipped_pos.sum()))
        assert (margins <= TIE_MARGIN).all(), margins.max()
        if head_l1 is not None:                                 # (the head's own planes: every first flip within the tie margin, every position off tolerance holds one)
            n_flip_h, margins_h, flipped_h = first_flip_margins(planes_to_dense(head_l1, 256), gold_spk, gold_vdec)
            assert (margins_h <= TIE_MARGIN).all(), margins_h.max()
            flipped_pos = flipped_h.any(axis=1)
            record_parity("rpn_head_full_size_counting_flips", level=lvl, flipped_neurons=n_flip_h, worst_margin=float(margins_h.max()) if n_flip_h else 0.0,
                          positions_off_tolerance=int(off_pos.size), positions_with_flip=int(flipped_pos.sum()))
        assert flipped_pos[off_pos].all(), "a position is off tolerance without any flipped hidden spike"
    if precision == "bf16x3":
        # the default launches (spike-rate outputs off: the structured-sparse conv, csrc/snn_sparse.h) at full size, attributed on the
        # spike planes the head itself left in its workspace (snn_debug_last_rpn_planes)
        import ctypes as Ct
        from snn_automotive_object_detection_amd import _lib, ops
        keep_counts = m.last_spike_counts
        m.spike_rates = False
        lg2, bb2 = m([f.to(gpu_device) for f in feats])
        assert _lib.load().snn_debug_last_conv_path() == 1
        off3 = (Ct.c_uint64 * 3)()
        _lib.load().snn_debug_last_rpn_planes(off3)
        P_all = int(off3[2])
        assert P_all == total
        ws = ops._WS.get(gpu_device, 1)
        raw = ws[int(off3[0]): int(off3[0]) + 8 * P_all * 8 * 4].view(torch.int32)
        planes = raw.view(8, 2, P_all, 4).permute(0, 2, 1, 3).reshape(8, P_all, 8) if off3[1] else raw.view(8, P_all, 8)
        bad2 = 0
        base = 0
        for l, (h, w) in enumerate(LEVELS):
            d2 = torch.maximum((lg2[l].cpu() - o_l[l]).abs().amax(dim=1), (bb2[l].cpu() - o_b[l]).abs().amax(dim=1)).reshape(-1).numpy()
            bad2 += int((d2 > 1e-4).sum())
            if l == 1:
                got = planes_to_dense(planes[:, base: base + 2 * h * w].contiguous(), 256)
                n_flip, margins, flipped = first_flip_margins(got, gold_spk, gold_vdec)
                assert float(got.mean()) > 0.001
                assert (margins <= TIE_MARGIN).all(), margins.max()
                assert flipped.any(axis=1)[np.nonzero(d2 > 1e-4)[0]].all(), "a position is off tolerance without any flipped hidden spike (sparse launch)"
                record_parity("rpn_head_full_size_sparse_flips", level=l, flipped_neurons=n_flip, worst_margin=float(margins.max()) if n_flip else 0.0,
                              positions_off_tolerance=int((d2 > 1e-4).sum()), positions_with_flip=int(flipped.any(axis=1).sum()))
            base += 2 * h * w
        record_parity("rpn_head_full_size_sparse", positions_off_tolerance=bad2, positions=total, budget=budget)
        assert bad2 <= budget, bad2
        m.spike_rates = True
        m.last_spike_counts = keep_counts
    # shared-LIF rates: the integer counts of the LIF epilogues against the oracle's own spike planes, per level and image
    counts = m.last_spike_counts.cpu().numpy()
    worst = 0
    for l, (h, w) in enumerate(LEVELS):
        n_neur = 8 * 256 * h * w
        gold = gold_counts[l].numpy()                                        # exact spike totals per image from the oracle
        diff = np.abs(counts[l, :2] - gold)
        worst = max(worst, int(diff.max()))
        assert (diff <= 4 * flip_budget(2 * h * w, 256, 8, "rpn_randn", precision)).all(), (l, counts[l, :2], gold)     # a flipped spike moves a count by a few
        r = rates[3 * l][:, 0].cpu().numpy()
        assert np.array_equal(r, (counts[l, :2].astype(np.float64) / n_neur).astype(np.float32))     # rate = count / (T*C*H*W)
    record_parity("rpn_head_full_size_counts", precision=precision, worst_count_difference=worst)


@pytest.mark.parametrize("precision", ["bf16x3", "f32", MXFP6_SWEEP])
def test_det_head_full_size_vs_oracle(gpu_device, monkeypatch, precision):
    import snn_automotive_object_detection_amd as S
    from oracle import snn_oracle as OR
    g = torch.Generator().manual_seed(8)
    x = torch.randn((2000, 256, 7, 7), generator=g)
    torch.manual_seed(1235)
    m = S.FastRCNNPredictorSNNFull(12544, 1024, 9, 12)
    with torch.no_grad():
        o_c, o_b, tr = OR.det_head_forward(x, m.fc6.weight, m.fc7.weight, m.cls_score.weight, m.bbox_pred.weight, 12, trace=True)
    tr = {k: tr[k] for k in ("cur6", "spk6", "cur7", "spk7")}
    m = m.to(gpu_device)
    m.precision = precision
    cls, bbox = m(x.to(gpu_device))
    d = torch.maximum((cls.cpu() - o_c).abs().amax(dim=1), (bbox.cpu() - o_b).abs().amax(dim=1))
    bad = int((d > 1e-4).sum())                                            # RoIs holding a flipped spike
    budget = flip_budget(2000, 2 * 1024, 12, "det", precision)             # two hidden layers of 1024 neurons, 12 steps
    record_parity("det_head_full_size", precision=precision, rois_off_tolerance=bad, rois=2000, budget=budget)
    assert bad <= budget, bad
    if precision == "bf16x3":
        # attribution on the hidden spike planes of the launches that actually ran (the head leaves them in its workspace:
        # snn_debug_last_det_planes; until round 4 a stage-level replica of the launches stood in, which the structured-sparse fc6 with its
        # own summation order no longer is): first flipped lif6 spikes sit on threshold ties; lif7 is checked on the RoIs whose lif6
        # trains equal the oracle's (teacher-forced by construction there); every RoI off tolerance holds a flipped spike
        import ctypes as Ct
        from snn_automotive_object_detection_amd import _lib, ops
        off3 = (Ct.c_uint64 * 3)()
        _lib.load().snn_debug_last_det_planes(off3)
        ws = ops._WS.get(gpu_device, 1)
        n_words = 12 * 2000 * 32
        p6 = ws[int(off3[0]): int(off3[0]) + 4 * n_words].view(torch.int32)
        p6 = p6.view(12, 32, 2000).permute(0, 2, 1).contiguous() if off3[2] else p6.view(12, 2000, 32)
        p7 = ws[int(off3[1]): int(off3[1]) + 4 * n_words].view(torch.int32).view(12, 2000, 32)
        g6, g7 = planes_to_dense(p6, 1024), planes_to_dense(p7, 1024)
        _, _, vdec6 = OR.lif_scan_from_currents(tr["cur6"])
        _, _, vdec7 = OR.lif_scan_from_currents(tr["cur7"])
        # (lif6's spikes of the last step are never read - dead time steps, csrc/snn_kernels.hip: det_windows - and not formed)
        n6, marg6, fl6 = first_flip_margins(g6[:11], tr["spk6"].numpy()[:11], vdec6.numpy()[:11])
        roi6 = fl6.any(axis=1)
        e7 = tr["spk7"].numpy()
        n7, marg7, fl7 = first_flip_margins(g7[:, ~roi6], e7[:, ~roi6], vdec7.numpy()[:, ~roi6])
        roi_any = roi6 | (g7 != e7).any(axis=(0, 2))
        record_parity("det_head_full_size_flips", lif6_flipped_neurons=n6, lif7_flipped_neurons_teacher_forced=n7,
                      worst_margin=float(max([0.0] + list(marg6) + list(marg7))), rois_off_tolerance=bad, rois_with_flip=int(roi_any.sum()))
        assert (marg6 <= TIE_MARGIN).all() and (marg7 <= TIE_MARGIN).all(), (marg6, marg7)
        assert float(g6.mean()) > 0.001 and float(g7.mean()) > 0.001            # (the planes read back are the head's)
        assert roi_any[(d > 1e-4).numpy()].all(), "a RoI is off tolerance without any flipped hidden spike"
    assert float(d.max()) < 0.1
    assert float(d.median()) < 1e-5


def test_full_size_properties(gpu_device):
    """determinism, image independence and the spike-count identity at the full pyramid"""
    import snn_automotive_object_detection_amd as S
    torch.manual_seed(3)
    m = S.RPNHeadSNN(256, 3, 8).to(gpu_device)
    feats = [torch.randn((2, 256, h, w), device=gpu_device) for h, w in LEVELS]
    a_l, a_b = m(feats)
    a_l = [t.clone() for t in a_l]
    b_l, b_b = m(feats)
    assert all(torch.equal(x, y) for x, y in zip(a_l, b_l))                # bitwise repeatable
    c_l, _ = m([f[1:2] for f in feats])                                    # image 1 alone
    assert all(torch.equal(x[1:2], y) for x, y in zip(a_l, c_l))
    z_l, z_b = m([torch.zeros_like(f) for f in feats])
    assert all(float(t.abs().max()) == 0.0 for t in z_l + z_b)


@pytest.mark.parametrize("precision", ["bf16x3", "f32", "mxfp6"])
def test_full_size_warm_repeats_are_bitwise_stable(gpu_device, precision):
    """six back-to-back calls of both heads at full size give the same bits: staging races show on warm repeats (caches hot,
    copies land early, the younger waves of a SIMD lag), not on a first call"""
    import snn_automotive_object_detection_amd as S
    torch.manual_seed(5)
    m = S.RPNHeadSNN(256, 3, 8).to(gpu_device)
    d = S.FastRCNNPredictorSNNFull(12544, 1024, 9, 12).to(gpu_device)
    m.precision = d.precision = precision
    feats = [torch.randn((2, 256, h, w), device=gpu_device) for h, w in LEVELS]
    x = torch.randn((2000, 256, 7, 7), device=gpu_device)
    first = None
    for _ in range(6):
        l, b = m(feats)
        c, r = d(x)
        cur = [t.clone() for t in l + b] + [c.clone(), r.clone()]
        if first is None:
            first = cur
        else:
            assert all(torch.equal(p, q) for p, q in zip(first, cur))


def test_bdd_shape_k11_vs_oracle(gpu_device):
    """BASELINE.json config[3] per-rank share at b=1: BDD 720x1280 -> 768x1376 canvas (odd level widths 43 and 22:
    partial tiles), K=11 classes"""
    import snn_automotive_object_detection_amd as S
    from oracle import snn_oracle as OR
    g = torch.Generator().manual_seed(11)
    levels = [(192, 344), (96, 172), (48, 86), (24, 43), (12, 22)]
    feats = [torch.randn((1, 256, h, w), generator=g) for h, w in levels]
    m = S.RPNHeadSNN(256, 3, 8)
    with torch.no_grad():
        o_l, o_b = OR.rpn_head_forward(feats, m.shared_conv.weight, m.conv_cls.weight, m.conv_bbox.weight, 8)
    m = m.to(gpu_device)
    for precision in ("bf16x3", "f32", "mxfp6"):
        m.precision = precision
        logits, bbox = m([f.to(gpu_device) for f in feats])
        total = bad = 0
        for l in range(5):
            d = torch.maximum((logits[l].cpu() - o_l[l]).abs().amax(1), (bbox[l].cpu() - o_b[l]).abs().amax(1))
            total += d.numel(); bad += int((d > 1e-4).sum())
        record_parity("bdd_rpn_head", precision=precision, positions_off_tolerance=bad, positions=total, budget=flip_budget(total, 256, 8, "rpn_randn", precision))
        assert bad <= flip_budget(total, 256, 8, "rpn_randn", precision), (precision, bad, total)
    x = torch.randn((300, 256, 7, 7), generator=g)
    d = S.FastRCNNPredictorSNNFull(12544, 1024, 11, 12)
    with torch.no_grad():
        o_c, o_d = OR.det_head_forward(x, d.fc6.weight, d.fc7.weight, d.cls_score.weight, d.bbox_pred.weight, 12)
    d = d.to(gpu_device)
    cls, box = d(x.to(gpu_device))
    assert tuple(cls.shape) == (300, 11) and tuple(box.shape) == (300, 44)
    off = torch.maximum((cls.cpu() - o_c).abs().amax(1), (box.cpu() - o_d).abs().amax(1))
    record_parity("bdd_det_head_k11", rois_off_tolerance=int((off > 1e-4).sum()), rois=300, budget=flip_budget(300, 2 * 1024, 12, "det"))
    assert int((off > 1e-4).sum()) <= flip_budget(300, 2 * 1024, 12, "det")


def test_bdd_config3_per_rank_size_vs_oracle(gpu_device):
    """BASELINE.json config[3] at its PER-RANK size (VERDICT r5 missing 1 / P-4): BDD 720x1280 -> 768x1376 canvas, **b = 4** images per GPU
    (/root/reference/configs/bdd.yaml:12, train.py:598-601), K = 11, 4 x 1000 = 4000 RoIs, bf16x3 - both heads free-running against the
    oracle, flips inside the usual budgets.  (~45 s of CPU oracle.)"""
    import snn_automotive_object_detection_amd as S
    from oracle import snn_oracle as OR
    g = torch.Generator().manual_seed(33)
    levels = [(192, 344), (96, 172), (48, 86), (24, 43), (12, 22)]
    feats = [torch.randn((4, 256, h, w), generator=g) for h, w in levels]
    torch.manual_seed(4321)
    m = S.RPNHeadSNN(256, 3, 8)
    with torch.no_grad():
        o_l, o_b = OR.rpn_head_forward(feats, m.shared_conv.weight, m.conv_cls.weight, m.conv_bbox.weight, 8)
    m = m.to(gpu_device)
    logits, bbox = m([f.to(gpu_device) for f in feats])
    from snn_automotive_object_detection_amd import _lib
    assert _lib.load().snn_debug_last_conv_path() == 1                       # (the structured-sparse launch, as in the bench's bdd leg)
    total = bad = 0
    for l in range(5):
        assert tuple(logits[l].shape) == (4, 3) + levels[l] and tuple(bbox[l].shape) == (4, 12) + levels[l]
        d = torch.maximum((logits[l].cpu() - o_l[l]).abs().amax(1), (bbox[l].cpu() - o_b[l]).abs().amax(1))
        total += d.numel(); bad += int((d > 1e-4).sum())
        assert float(d.max()) < 0.05
    assert total == 4 * 87984
    budget = flip_budget(total, 256, 8, "rpn_randn")
    record_parity("bdd_config3_b4_rpn_head", positions_off_tolerance=bad, positions=total, budget=budget)
    assert bad <= budget, (bad, total)
    del feats, o_l, o_b, logits, bbox
    x = torch.randn((4000, 256, 7, 7), generator=g)
    dh = S.FastRCNNPredictorSNNFull(12544, 1024, 11, 12)
    with torch.no_grad():
        o_c, o_d = OR.det_head_forward(x, dh.fc6.weight, dh.fc7.weight, dh.cls_score.weight, dh.bbox_pred.weight, 12)
    dh = dh.to(gpu_device)
    cls, box = dh(x.to(gpu_device))
    assert _lib.load().snn_debug_last_fc6_path() == 1
    assert tuple(cls.shape) == (4000, 11) and tuple(box.shape) == (4000, 44)
    off = torch.maximum((cls.cpu() - o_c).abs().amax(1), (box.cpu() - o_d).abs().amax(1))
    n_off = int((off > 1e-4).sum())
    budget = flip_budget(4000, 2 * 1024, 12, "det")
    record_parity("bdd_config3_b4_det_head_k11", rois_off_tolerance=n_off, rois=4000, budget=budget)
    assert n_off <= budget and float(off.max()) < 0.5, (n_off, float(off.max()))


def test_stress_config_T16_T24_with_spike_rates(gpu_device):
    """BASELINE.json config[4] at a reduced canvas: T_rpn=16 / T_det=24 with the spike-rate outputs on"""
    import snn_automotive_object_detection_amd as S
    from oracle import snn_oracle as OR
    g = torch.Generator().manual_seed(12)
    feats = [torch.randn((2, 256, 24, 48), generator=g), torch.randn((2, 256, 12, 24), generator=g)]
    m = S.RPNHeadSNN(256, 3, 16)
    gc = []
    with torch.no_grad():
        o_l, o_b, o_r = OR.rpn_head_forward(feats, m.shared_conv.weight, m.conv_cls.weight, m.conv_bbox.weight, 16,
                                            spike_rates=True, counts_out=gc)
    m = m.to(gpu_device)
    m.spike_rates = True
    logits, bbox, rates = m([f.to(gpu_device) for f in feats])
    assert len(rates) == 6
    counts = m.last_spike_counts.cpu().numpy()
    for l, (h, w) in enumerate([(24, 48), (12, 24)]):          # shared-LIF rates: integer counts / (T*C*H*W)
        n_neur = 16 * 256 * h * w
        gold = gc[l].numpy()
        assert (np.abs(counts[l] - gold) <= 4 * flip_budget(2 * h * w, 256, 16)).all(), (counts[l], gold)
        assert np.array_equal(rates[3 * l][:, 0].cpu().numpy(), (counts[l].astype(np.float64) / n_neur).astype(np.float32))
        assert torch.equal(rates[3 * l][:, 1].cpu(), o_r[3 * l][:, 1])
    bad = sum(int(((logits[l].cpu() - o_l[l]).abs().amax(1) > 1e-4).sum()) for l in range(2))
    assert bad <= flip_budget(2 * (24 * 48 + 12 * 24), 256, 16)
    x = torch.randn((64, 256, 7, 7), generator=g)
    d = S.FastRCNNPredictorSNNFull(12544, 1024, 9, 24)
    gd = []
    with torch.no_grad():
        o = OR.det_head_forward(x, d.fc6.weight, d.fc7.weight, d.cls_score.weight, d.bbox_pred.weight, 24, spike_rates=True,
                                counts_out=gd)
    d = d.to(gpu_device)
    d.spike_rates = True
    r = d(x.to(gpu_device))
    assert len(r) == 4 and all(tuple(t.shape) == (64, 2) for t in r)
    c6, c7 = [c.cpu().numpy() for c in d.last_spike_counts]
    for j, c in enumerate((c6, c7)):                           # per-RoI spike counts of lif6 / lif7: integers, equal to the oracle's
        gold = gd[j].numpy()
        assert int((c != gold).sum()) <= flip_budget(64, 1024, 24), (j, int((c != gold).sum()))
        assert np.array_equal(r[j][:, 0].cpu().numpy(), (c.astype(np.float64) / (24 * 1024)).astype(np.float32))
        assert torch.equal(r[j][:, 1].cpu(), o[j][:, 1])
    same = torch.from_numpy((c6 == gd[0].numpy()) & (c7 == gd[1].numpy()))      # RoIs whose hidden spike counts equal the oracle's
    assert int((~same).sum()) <= flip_budget(64, 2 * 1024, 24, "det")
    for j in (2, 3):                                           # LI "rates": means of membrane sums (a flipped spike moves them)
        assert torch.allclose(r[j].cpu()[same], o[j][same], rtol=1e-4, atol=2e-5)


def test_heads_are_hipgraph_capturable(gpu_device):
    """include/snn_hip.h promises stream-only work: capture both heads in a HIP graph and replay"""
    import snn_automotive_object_detection_amd as S
    torch.manual_seed(5)
    rpn = S.RPNHeadSNN(64, 3, 10).to(gpu_device)
    det = S.FastRCNNPredictorSNNFull(16 * 49, 128, 5, 12).to(gpu_device)
    with torch.no_grad():
        rpn.shared_conv.weight.mul_(5.0)                        # make sure spikes reach the outputs
    f = [torch.randn((1, 64, 16, 24), device=gpu_device) * 2]
    x = torch.randn((50, 16, 7, 7), device=gpu_device) * 2
    ref_l, ref_b = rpn(f)
    ref_c, ref_d = det(x)                                      # warm-up: packs weights, sizes the workspace
    ref_l0, ref_c0 = ref_l[0].clone(), ref_c.clone()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        g_l, g_b = rpn(f)
        g_c, g_d = det(x)
    f[0].copy_(torch.randn_like(f[0]) * 2)                     # new inputs, same buffers
    x.copy_(torch.randn_like(x) * 2)
    graph.replay()
    torch.cuda.synchronize()
    exp_l, _ = rpn(f)
    exp_c, _ = det(x)
    assert torch.equal(g_l[0], exp_l[0]) and torch.equal(g_c, exp_c)
    assert float(exp_l[0].abs().max()) > 0 and float(exp_c.abs().max()) > 0
    assert not torch.equal(g_l[0], ref_l0) and not torch.equal(g_c, ref_c0)


def test_stress_config_full_canvas_T16_T24(gpu_device):
    """BASELINE.json config[4] at FULL size: T_rpn=16 on the whole Cityscapes pyramid (b=2), T_det=24 on 2000 RoIs, spike-rate
    outputs on - logits / deltas / integer spike counts against the oracle"""
    import snn_automotive_object_detection_amd as S
    from oracle import snn_oracle as OR
    g = torch.Generator().manual_seed(21)
    feats = [torch.randn((2, 256, h, w), generator=g) for h, w in LEVELS]
    m = S.RPNHeadSNN(256, 3, 16)
    gc = []
    with torch.no_grad():
        o_l, o_b, o_r = OR.rpn_head_forward(feats, m.shared_conv.weight, m.conv_cls.weight, m.conv_bbox.weight, 16, spike_rates=True,
                                            counts_out=gc)
    m = m.to(gpu_device)
    m.spike_rates = True
    logits, bbox, rates = m([f.to(gpu_device) for f in feats])
    total = bad = 0
    for l in range(5):
        d = torch.maximum((logits[l].cpu() - o_l[l]).abs().amax(1), (bbox[l].cpu() - o_b[l]).abs().amax(1))
        total += d.numel(); bad += int((d > 1e-4).sum())
    record_parity("stress_rpn_full_T16", positions_off_tolerance=bad, positions=total, budget=flip_budget(total, 256, 16, "rpn_randn"))
    assert bad <= flip_budget(total, 256, 16, "rpn_randn"), (bad, total)
    counts = m.last_spike_counts.cpu().numpy()
    for l, (h, w) in enumerate(LEVELS):
        n_neur = 16 * 256 * h * w
        gold = gc[l].numpy()
        assert (np.abs(counts[l] - gold) <= 4 * flip_budget(2 * h * w, 256, 16)).all(), (l, counts[l], gold)
        for j in (1, 2):                                       # LI "rates" (means of membrane sums) and the FLOP constants
            assert torch.allclose(rates[3 * l + j].cpu(), o_r[3 * l + j], rtol=1e-4, atol=2e-5)
        assert torch.equal(rates[3 * l][:, 1].cpu(), o_r[3 * l][:, 1])
    x = torch.randn((2000, 256, 7, 7), generator=g)
    d = S.FastRCNNPredictorSNNFull(12544, 1024, 9, 24)
    with torch.no_grad():
        gd = []
        o_c, o_d = OR.det_head_forward(x, d.fc6.weight, d.fc7.weight, d.cls_score.weight, d.bbox_pred.weight, 24, counts_out=gd)
    d = d.to(gpu_device)
    cls, box = d(x.to(gpu_device))
    off = torch.maximum((cls.cpu() - o_c).abs().amax(1), (box.cpu() - o_d).abs().amax(1))
    n_off = int((off > 1e-4).sum())
    record_parity("stress_det_full_T24", rois_off_tolerance=n_off, rois=2000, budget=flip_budget(2000, 2 * 1024, 24, "det"))
    assert n_off <= flip_budget(2000, 2 * 1024, 24, "det"), n_off
    d.spike_rates = True
    r = d(x.to(gpu_device))
    c6, c7 = [c.cpu().numpy() for c in d.last_spike_counts]
    for j, c in enumerate((c6, c7)):
        gold = gd[j].numpy()
        assert int((c != gold).sum()) <= flip_budget(2000, 1024 * (j + 1), 24), (j, int((c != gold).sum()))
    assert len(r) == 4 and all(tuple(t.shape) == (2000, 2) for t in r)
