"""GPU parity of the drop-in modules (RPNHeadSNN / FastRCNNPredictorSNNFull) against the committed
golden fixtures (outputs of the reference's own forward bodies run under shims).

Free-running tolerance (north_star: 1e-4 fp32) with the flip budget of SURVEY.md §7 risk 1:
outputs must match within 1e-4 at every position (RPN) / RoI (detector) whose hidden spike trains
are identical to the golden ones; a spike may only differ where a threshold tie makes the fp32
summation order decide, and such positions are counted against a budget."""
import numpy as np
import pytest
import torch

from oracle import fixtures as FX
from tests._util import planes_to_dense, flip_budget, flip_lambda

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def pkg():
    import snn_automotive_object_detection_amd as pkg
    return pkg


PRECISIONS = ["bf16x3", "f32", "mxfp6"]     # mxfp6 falls back to bf16x3 where C % 128 != 0


def _rpn_module(pkg, spec, dev, feats_w, precision="bf16x3"):
    feats, w_s, w_c, w_b = feats_w
    m = pkg.RPNHeadSNN(spec["C"], spec["A"], spec["T"]).to(dev)
    m.precision = precision
    m.load_state_dict({"shared_conv.weight": w_s, "conv_cls.weight": w_c, "conv_bbox.weight": w_b})
    return m


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("name", sorted(FX.RPN_SPECS))
def test_rpn_head_vs_golden(pkg, gpu_device, name, precision):
    from snn_automotive_object_detection_amd import ops
    spec = FX.RPN_SPECS[name]
    exp = FX.load_expected(name)
    inp = FX.rpn_inputs(spec)
    m = _rpn_module(pkg, spec, gpu_device, inp, precision)
    feats = [f.to(gpu_device) for f in inp[0]]
    logits, bbox = m(feats)
    assert len(logits) == len(feats) and len(bbox) == len(feats)
    # hidden spikes of our own run (stage ops) to locate flipped positions
    p = m._params()
    n_bad_total = 0
    for l, f in enumerate(feats):
        N, C, H, W = f.shape
        assert tuple(logits[l].shape) == (N, spec["A"], H, W)
        assert tuple(bbox[l].shape) == (N, 4 * spec["A"], H, W)
        enc = ops.encode_nchw(f, spec["T"], p)
        if precision == "f32":
            spk = ops.conv3x3_lif(enc, N, C, C, H, W, p, ops.pack_conv3x3(m.shared_conv.weight))
        else:
            spk = ops.conv3x3_lif_bf16x3(enc, [(N, H, W)], C, C, p, ops.pack_conv3x3_bf16x3(m.shared_conv.weight))
        got = planes_to_dense(spk, C).reshape(spec["T"], N, H, W, C)
        gold = FX.unpack_spikes(exp["spk%d" % l], exp["spk%d_shape" % l]).transpose(0, 1, 3, 4, 2)
        flipped_pos = (got != gold).any(axis=(0, 4))                       # [N,H,W]
        n_bad_total += int(flipped_pos.sum())
        ok = ~flipped_pos[:, None, :, :]
        for o, key in ((logits[l], "logits%d" % l), (bbox[l], "bbox%d" % l)):
            d = np.abs(o.cpu().numpy() - exp[key])
            assert (d * ok).max() <= TOL, (key, float((d * ok).max()))
            assert d.max() < 0.05            # a flipped spike moves an output by ~1e-3, never by much
    total_pos = sum(f.shape[0] * f.shape[2] * f.shape[3] for f in feats)
    assert n_bad_total <= flip_budget(total_pos, spec["C"], spec["T"]), "flipped positions: %d of %d" % (n_bad_total, total_pos)


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("name", sorted(FX.RPN_SPECS))
def test_rpn_head_spike_rates_vs_golden(pkg, gpu_device, name, precision):
    spec = FX.RPN_SPECS[name]
    exp = FX.load_expected(name)
    inp = FX.rpn_inputs(spec)
    m = _rpn_module(pkg, spec, gpu_device, inp, precision)
    m.spike_rates = True
    logits, bbox, rates = m([f.to(gpu_device) for f in inp[0]])
    assert len(rates) == 3 * len(inp[0])
    for l in range(len(inp[0])):
        for j in range(3):
            r = rates[3 * l + j].cpu().numpy()
            e = exp["rate%d_%d" % (l, j)]
            assert r.dtype == np.float32 and r.shape == e.shape
            assert np.array_equal(r[:, 1], e[:, 1])                          # "FLOPs" column: exact
            if j == 0:
                # shared LIF: the integer spike count of the fused epilogue against the count of the golden spike planes
                T, C = spec["T"], spec["C"]
                N, _, H, W = inp[0][l].shape
                gold = FX.unpack_spikes(exp["spk%d" % l], exp["spk%d_shape" % l]).reshape(T, N, -1).sum(axis=(0, 2))
                cnt = m.last_spike_counts[l, :N].cpu().numpy()
                assert (np.abs(cnt - gold) <= 12 * flip_lambda(N * H * W, C, T)).all(), (l, cnt, gold)   # equal unless a spike flipped (fixture sizes: lambda << 0.1)
                assert np.array_equal(r[:, 0], (cnt.astype(np.float64) / (T * C * H * W)).astype(np.float32))
                np.testing.assert_allclose(r[:, 0], e[:, 0], rtol=3e-7 + 4e-3 * (cnt != gold).any(), atol=0)
            else:
                # means of signed LI membranes (they cancel, so the bound is absolute: one flipped hidden spike moves such a
                # mean by ~1e-3/(A*H*W))
                np.testing.assert_allclose(r[:, 0], e[:, 0], rtol=1e-4, atol=2e-5)
        d = np.abs(logits[l].cpu().numpy() - exp["logits%d" % l])
        assert (d > TOL).sum() <= 3 * spec["A"]        # at most a few flipped positions


def _det_module(pkg, spec, dev, inp, precision="bf16x3"):
    x, w6, w7, wc, wb = inp
    m = pkg.FastRCNNPredictorSNNFull(spec["C"] * 49, spec["Hd"], spec["K"], spec["T"],
                                     only_one_bbox=spec.get("only_one_bbox", False)).to(dev)
    m.precision = precision
    m.load_state_dict({"fc6.weight": w6, "fc7.weight": w7, "cls_score.weight": wc, "bbox_pred.weight": wb})
    return m


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("name", sorted(FX.DET_SPECS))
def test_det_head_vs_golden(pkg, gpu_device, name, precision):
    from snn_automotive_object_detection_amd import ops
    spec = FX.DET_SPECS[name]
    exp = FX.load_expected(name)
    inp = FX.det_inputs(spec)
    m = _det_module(pkg, spec, gpu_device, inp, precision)
    x = inp[0].to(gpu_device)
    cls, bbox = m(x)
    R, T, Hd, D = x.shape[0], spec["T"], spec["Hd"], spec["C"] * 49
    assert tuple(cls.shape) == exp["cls"].shape and tuple(bbox.shape) == exp["bbox"].shape
    # our own hidden spikes, free-running, to find RoIs with a flipped spike
    p = m._params()
    enc = ops.encode_rows(x.flatten(1), T, p)
    gemm, pack = (ops.spike_gemm, ops.pack_linear) if precision == "f32" else (ops.spike_gemm_bf16x3, ops.pack_linear_bf16x3)
    cur6 = gemm(enc.view(T * R, -1), D, Hd, pack(m.fc6.weight)).view(T, R, -1)
    s6 = ops.lif_scan(cur6, Hd, p)
    cur7 = gemm(s6.view(T * R, -1), Hd, Hd, pack(m.fc7.weight)).view(T, R, -1)
    s7 = ops.lif_scan(cur7, Hd, p)
    g6 = FX.unpack_spikes(exp["spk6"], exp["spk6_shape"])
    g7 = FX.unpack_spikes(exp["spk7"], exp["spk7_shape"])
    bad = (planes_to_dense(s6, Hd) != g6).any(axis=(0, 2)) | (planes_to_dense(s7, Hd) != g7).any(axis=(0, 2))
    assert bad.sum() <= flip_budget(R, 2 * Hd, T) - 1, "RoIs with flipped spikes: %d of %d" % (bad.sum(), R)
    ok = ~bad[:, None]
    for o, key in ((cls, "cls"), (bbox, "bbox")):
        d = np.abs(o.cpu().numpy() - exp[key])
        assert (d * ok).max() <= TOL, (key, float((d * ok).max()))
        assert d.max() < 0.1


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("name", sorted(FX.DET_SPECS))
def test_det_head_spike_rates_vs_golden(pkg, gpu_device, name, precision):
    spec = FX.DET_SPECS[name]
    exp = FX.load_expected(name)
    inp = FX.det_inputs(spec)
    m = _det_module(pkg, spec, gpu_device, inp, precision)
    m.spike_rates = True
    rates = m(inp[0].to(gpu_device))
    assert isinstance(rates, list) and len(rates) == 4
    T, Hd, R = spec["T"], spec["Hd"], spec["R"]
    golds = [FX.unpack_spikes(exp[k], exp[k + "_shape"]).sum(axis=(0, 2)) for k in ("spk6", "spk7")]     # spikes per RoI
    for j, r in enumerate(rates):
        r = r.cpu().numpy(); e = exp["rate%d" % j]
        assert r.dtype == np.float32 and r.shape == e.shape
        assert np.array_equal(r[:, 1], e[:, 1])
        if j < 2:        # lif6 / lif7: integer counts from the LIF epilogues == counts of the golden planes (flipped RoIs aside)
            cnt = m.last_spike_counts[j].cpu().numpy()
            assert int((cnt != golds[j]).sum()) <= flip_budget(R, Hd * (j + 1), T) - 1, (j, int((cnt != golds[j]).sum()))
            assert np.array_equal(r[:, 0], (cnt.astype(np.float64) / (T * Hd)).astype(np.float32))
            same = cnt == golds[j]
            np.testing.assert_allclose(r[same, 0], e[same, 0], rtol=3e-7, atol=0)
        else:
            bad = np.abs(r[:, 0] - e[:, 0]) > (1e-4 * np.abs(e[:, 0]) + 2e-5)
            assert bad.sum() <= flip_budget(R, 2 * Hd, T) - 1


# ---------------------------------------------------------------------------------------------
# edge cases and error behaviour
# ---------------------------------------------------------------------------------------------
def test_det_head_empty_and_single_roi(pkg, gpu_device):
    m = pkg.FastRCNNPredictorSNNFull(8 * 49, 64, 5, 6).to(gpu_device)
    c, b = m(torch.zeros((0, 8, 7, 7), device=gpu_device))
    assert tuple(c.shape) == (0, 5) and tuple(b.shape) == (0, 20)
    c, b = m(torch.randn((1, 8, 7, 7), device=gpu_device))
    assert tuple(c.shape) == (1, 5) and torch.isfinite(c).all() and torch.isfinite(b).all()


def test_zero_and_subthreshold_input_gives_exact_zero(pkg, gpu_device):
    m = pkg.RPNHeadSNN(64, 3, 8).to(gpu_device)
    f = torch.full((1, 64, 9, 11), 0.3, device=gpu_device)   # never crosses 0.25 within 8 steps
    l, b = m([f, torch.zeros((1, 64, 3, 3), device=gpu_device)])
    assert all(float(t.abs().max()) == 0.0 for t in l + b)


def test_errors_are_loud(pkg, gpu_device):
    from snn_automotive_object_detection_amd._lib import SnnHipError
    m = pkg.RPNHeadSNN(32, 3, 8).to(gpu_device)
    with pytest.raises(SnnHipError):
        m([torch.randn(1, 32, 4, 4)])                         # CPU tensor: no fallback
    with pytest.raises(SnnHipError):
        m([torch.randn(1, 16, 4, 4, device=gpu_device)])      # wrong channel count
    m33 = pkg.RPNHeadSNN(32, 3, 33).to(gpu_device)
    with pytest.raises(SnnHipError):
        m33([torch.randn(1, 32, 4, 4, device=gpu_device)])    # more steps than SNN_MAX_STEPS


def test_non_contiguous_and_half_inputs(pkg, gpu_device):
    torch.manual_seed(0)
    m = pkg.RPNHeadSNN(32, 3, 6).to(gpu_device)
    f = torch.randn(2, 32, 10, 12, device=gpu_device)
    ref_l, ref_b = m([f])
    cl_l, cl_b = m([f.to(memory_format=torch.channels_last)])
    assert torch.equal(ref_l[0], cl_l[0]) and torch.equal(ref_b[0], cl_b[0])
    h = f.half()
    h_l, _ = m([h])
    e_l, _ = m([h.float()])
    assert torch.equal(h_l[0], e_l[0])


def test_batch_and_level_independence_bitwise(pkg, gpu_device):
    """images and levels never interact: any regrouping gives bit-identical per-image outputs"""
    torch.manual_seed(1)
    m = pkg.RPNHeadSNN(64, 3, 8).to(gpu_device)
    f0 = torch.randn(3, 64, 13, 9, device=gpu_device)
    f1 = torch.randn(3, 64, 5, 20, device=gpu_device)
    l_all, b_all = m([f0, f1])
    l_0, b_0 = m([f0])
    l_1, b_1 = m([f1[1:2]])
    assert torch.equal(l_all[0], l_0[0]) and torch.equal(b_all[0], b_0[0])
    assert torch.equal(l_all[1][1:2], l_1[0]) and torch.equal(b_all[1][1:2], b_1[0])
    l_p, _ = m([f0.flip(0)])
    assert torch.equal(l_p[0].flip(0), l_0[0])


def test_roi_permutation_equivariance_bitwise(pkg, gpu_device):
    torch.manual_seed(2)
    m = pkg.FastRCNNPredictorSNNFull(16 * 49, 128, 7, 12).to(gpu_device)
    x = torch.randn(150, 16, 7, 7, device=gpu_device)
    perm = torch.randperm(150, device=gpu_device)
    c, b = m(x)
    cp, bp = m(x[perm])
    assert torch.equal(c[perm], cp) and torch.equal(b[perm], bp)


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("C,T,shape", [(320, 3, (1, 9, 14)), (96, 1, (2, 8, 8)), (32, 32, (1, 5, 9)), (384, 2, (1, 8, 8))])
def test_rpn_head_odd_widths_and_step_limits_vs_oracle(pkg, gpu_device, precision, C, T, shape):
    """channel counts that are not 256 (odd number of 32-channel words, the f32 kernel's 384 maximum), T = 1 and
    T = SNN_MAX_STEPS, against the oracle run on the spot"""
    from oracle import snn_oracle as OR
    g = torch.Generator().manual_seed(C + T)
    m = pkg.RPNHeadSNN(C, 3, T)
    f = torch.randn((shape[0], C, shape[1], shape[2]), generator=g) * 1.5
    with torch.no_grad():
        o_l, o_b = OR.rpn_head_forward([f], m.shared_conv.weight, m.conv_cls.weight, m.conv_bbox.weight, T)
    m = m.to(gpu_device)
    m.precision = precision
    l, b = m([f.to(gpu_device)])
    d = torch.maximum((l[0].cpu() - o_l[0]).abs().amax(1), (b[0].cpu() - o_b[0]).abs().amax(1))
    assert int((d > TOL).sum()) <= 2 and float(d.max()) < 0.05


def test_f32_conv_rejects_too_many_channels_but_bf16x3_takes_them(pkg, gpu_device):
    from snn_automotive_object_detection_amd._lib import SnnHipError
    m = pkg.RPNHeadSNN(416, 3, 2).to(gpu_device)
    f = torch.randn((1, 416, 8, 8), device=gpu_device)
    m.precision = "f32"
    with pytest.raises(SnnHipError):
        m([f])                                   # fp32 LDS spike image: C_in <= 384
    m.precision = "bf16x3"
    l, b = m([f])
    assert tuple(l[0].shape) == (1, 3, 8, 8) and torch.isfinite(l[0]).all()


def test_two_host_threads_two_streams_bitwise(pkg, gpu_device):
    """the library is re-entrant (snn_hip.h): two host threads, each on its own HIP stream (workspaces are per stream),
    run both heads concurrently and reproduce the single-stream results bit for bit"""
    import threading
    torch.manual_seed(3)
    rpn = pkg.RPNHeadSNN(128, 3, 8).to(gpu_device)
    det = pkg.FastRCNNPredictorSNNFull(32 * 49, 256, 9, 12).to(gpu_device)
    feats = [[torch.randn(2, 128, 40 + 8 * i, 56, device=gpu_device), torch.randn(2, 128, 20, 28, device=gpu_device)] for i in range(2)]
    rois = [torch.randn(300 + 50 * i, 32, 7, 7, device=gpu_device) for i in range(2)]
    with torch.no_grad():
        ref = [(rpn(feats[i]), det(rois[i])) for i in range(2)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(gpu_device) for _ in range(2)]
    bad, errs = [], []

    def worker(i):
        try:
            with torch.no_grad(), torch.cuda.stream(streams[i]):
                fi = [f.clone() for f in feats[i]]
                ri = rois[i].clone()
                for it in range(12):
                    (lg, bb), (c, b) = rpn(fi), det(ri)
                    (rl, rb), (rc, rbx) = ref[i]
                    same = all(torch.equal(x, y) for x, y in zip(lg + bb, rl + rb)) and torch.equal(c, rc) and torch.equal(b, rbx)
                    if not same:
                        bad.append((i, it))
                streams[i].synchronize()
        except Exception as e:                      # surfaced in the main thread
            errs.append(repr(e))
    th = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    assert not errs, errs
    assert not bad, bad


def test_stream_pipeline_keeps_order_and_values(pkg, gpu_device):
    """StreamPipeline: batches in flight on two host threads / HIP streams come back in order with the single-stream values;
    an exception inside a slot reaches the caller"""
    torch.manual_seed(4)
    rpn = pkg.RPNHeadSNN(64, 3, 8).to(gpu_device)
    batches = [[torch.randn(1, 64, 20 + i, 30, device=gpu_device)] for i in range(7)]
    with torch.no_grad():
        ref = [rpn(b) for b in batches]
    got = pkg.StreamPipeline(rpn, slots=2).map(batches)
    assert len(got) == len(ref)
    for (rl, rb), (gl, gb) in zip(ref, got):
        assert torch.equal(rl[0], gl[0]) and torch.equal(rb[0], gb[0])
    assert pkg.StreamPipeline(rpn, slots=3).map([]) == []

    def boom(b):
        raise RuntimeError("slot failure")
    with pytest.raises(RuntimeError, match="slot failure"):
        pkg.StreamPipeline(boom, slots=2).map(batches)


def test_stream_pipeline_on_a_cold_model(pkg, gpu_device):
    """a freshly constructed model handed to StreamPipeline.map - the first batches of the slots fill the packed-weight
    / folded-BatchNorm / anchor caches concurrently on different streams.  Results must equal the warmed single-stream ones.
    (Whole model at a small canvas: the stock backbone is not bitwise repeatable, so its features are computed once and the
    pipeline runs everything behind it; the cold FrozenBatchNorm path is exercised by a cold backbone block.)"""
    import snn_automotive_object_detection_amd as S
    from snn_automotive_object_detection_amd.stock.backbone import Bottleneck
    torch.manual_seed(6)
    for trial in range(3):
        rpn = pkg.RPNHeadSNN(64, 3, 8).to(gpu_device)
        det = pkg.FastRCNNPredictorSNNFull(16 * 49, 128, 5, 12).to(gpu_device)
        blk = Bottleneck(64, 16).to(gpu_device).eval()                     # 3 FrozenBatchNorm2d with cold folded constants
        for bn in (blk.bn1, blk.bn2, blk.bn3):
            bn.running_var.uniform_(0.5, 2.0); bn.weight.uniform_(0.5, 1.5); bn.bias.normal_()

        def model(b):
            f, x = b
            y = blk(f[0])
            return rpn([y]), det(x)
        batches = [([torch.randn(1, 64, 24, 30 + i, device=gpu_device)], torch.randn(40 + i, 16, 7, 7, device=gpu_device)) for i in range(6)]
        torch.cuda.synchronize()
        got = pkg.StreamPipeline(model, slots=3).map(batches)              # COLD: nothing has run on these modules yet
        with torch.no_grad():
            ref = [model(b) for b in batches]                              # warmed, one stream
        for ((rl, rb), (rc, rd)), ((gl, gb), (gc, gd)) in zip(ref, got):
            assert torch.equal(rl[0], gl[0]) and torch.equal(rb[0], gb[0]) and torch.equal(rc, gc) and torch.equal(rd, gd), trial
