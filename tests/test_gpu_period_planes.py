"""Period planes (csrc/snn_common.h).  The constant-current encoder starts from and resets to +0, so its spike train is exactly
periodic: z_t = 1 iff n | t + 1 with n = (first spike step) + 1.  The bf16x3 heads multiply the DISJOINT planes e_n = (period == n)
instead of the spike planes z_t (a quarter of the operand switching in the power-limited matrix-core loop) and add up the
divisors' partial currents in the LIF epilogue: cur_t = sum over n | t + 1 of W e_n.  Checked here:
  * the period planes the encoders emit reconstruct the spike planes bit for bit (z_t = OR over n | t+1 of e_n) and are disjoint;
  * stage level, teacher-forced: conv+LIF / linear+LIF on period planes against the oracle's spikes, every flip on a threshold tie;
  * whole heads: period mode (default) against spike-plane mode (SNN_PERIOD_PLANES=0) - two fp32 summation orders of the same
    currents - differ in at most a flip budget of positions / RoIs; each is checked against the oracle by the ordinary tests."""
import numpy as np
import pytest
import torch

from oracle import fixtures as FX
from oracle import snn_oracle as OR
from tests._util import TIE_MARGIN, dense_to_planes, first_flip_margins, flip_budget, nchw_to_rows, planes_to_dense

pytestmark = pytest.mark.gpu


def _params(ops):
    return ops.make_params(ops.LIFParameters(v_th=torch.tensor(0.25)), ops.LIFParameters(alpha=100, v_th=torch.tensor(0.1)))


def _reconstruct(e):
    """period planes [T, rows, words] (int32) -> spike planes"""
    T = e.shape[0]
    z = torch.zeros_like(e)
    for t in range(T):
        for n in range(1, T + 1):
            if (t + 1) % n == 0:
                z[t] |= e[n - 1]
    return z


@pytest.mark.parametrize("T", [1, 2, 7, 12, 24])
def test_encoders_emit_disjoint_period_planes_that_reconstruct_the_spike_planes(gpu_device, monkeypatch, T):
    from snn_automotive_object_detection_amd import ops
    from snn_automotive_object_detection_amd.stock.roi_align import MultiScaleRoIAlign
    p = _params(ops)
    g = torch.Generator().manual_seed(T)
    f = (torch.randn(2, 70, 13, 17, generator=g) * 3.0).to(gpu_device)
    x = (torch.randn(37, 320, generator=g) * 3.0).to(gpu_device)
    x[0, :6] = torch.tensor([2.5, 2.4999998, 2.5000002, 0.25, 0.26, 100.0])        # threshold neighbours, a period-1 neuron
    # every threshold of the encoder's table and its neighbours two ulps either side (csrc/snn_common.h: THRESHOLD FORM)
    import ctypes as C
    from snn_automotive_object_detection_amd import _lib
    th = (C.c_float * 32)()
    assert _lib.load().snn_debug_encoder_thresholds(C.byref(p), th) == 1
    bits = np.array(list(th), dtype=np.float32).view(np.uint32)
    near = np.concatenate([(b + np.arange(-2, 3)).astype(np.uint32).view(np.float32) for b in bits])       # 160 values
    x[1, :160] = torch.from_numpy(near.copy()).to(gpu_device)
    f[0, :40, 0, :4] = torch.from_numpy(near.copy()).reshape(40, 4).to(gpu_device)
    feats = {str(i): (torch.randn(2, 8, 24 >> i, 40 >> i, generator=g) * 3).to(gpu_device) for i in range(4)}
    boxes = [torch.tensor([[3.0, 4.0, 60.0, 50.0], [10.0, 10.0, 150.0, 90.0], [0.0, 0.0, 20.0, 12.0]], device=gpu_device)] * 2
    pool = MultiScaleRoIAlign(["0", "1", "2", "3"], 7, 2)
    flist, scales, rois, lvl = pool.assign(feats, boxes, [(96, 160)] * 2)

    def all_planes():
        return [ops.encode_nchw(f, T, p), ops.encode_rows(x, T, p), ops.roi_align_encode(flist, scales, rois[:, 1:5], rois[:, 0], lvl, T, p)]
    direct = all_planes()
    monkeypatch.setenv("SNN_STAGE_PERIODS", "1")
    periods = all_planes()
    monkeypatch.setenv("SNN_ENC_ROWS", "ballot")
    periods.append(ops.encode_rows(x, T, p)); direct.append(direct[1])
    monkeypatch.delenv("SNN_ENC_ROWS")
    monkeypatch.setenv("SNN_ENC_QUANT", "0")                                        # period planes by the recurrence: the same bits
    for a, b in zip(all_planes(), periods[:3]):
        assert torch.equal(a, b)
    for z, e in zip(direct, periods):
        acc = torch.zeros_like(e[0])
        for t in range(T):
            assert int((acc & e[t]).ne(0).sum()) == 0                               # disjoint: a neuron has ONE first spike
            acc |= e[t]
        assert torch.equal(_reconstruct(e), z)
        assert torch.equal(e[0], z[0])
    assert int(direct[0].ne(0).sum()) > 0


@pytest.mark.parametrize("name", ["rpn_c256_T8_odd", "rpn_c64_A5_T8", "rpn_c256_T12", "rpn_c256_T4"])
def test_conv3x3_lif_on_period_planes_teacher_forced(gpu_device, monkeypatch, name):
    from snn_automotive_object_detection_amd import ops
    spec = FX.RPN_SPECS[name]
    T, C = spec["T"], spec["C"]
    feats, w_s, w_c, w_b = FX.rpn_inputs(spec)
    p = _params(ops)
    wp = ops.pack_conv3x3_bf16x3(w_s.to(gpu_device))
    _, _, traces = OR.rpn_head_forward(feats, w_s, w_c, w_b, T, trace=True)
    monkeypatch.setenv("SNN_STAGE_PERIODS", "1")
    for f, tr in zip(feats, traces):
        N, _, H, W = f.shape
        enc = ops.encode_nchw(f.to(gpu_device), T, p)                              # period planes
        assert torch.equal(_reconstruct(enc).cpu(), dense_to_planes(nchw_to_rows(tr["z"])))
        spk = ops.conv3x3_lif_bf16x3(enc, [(N, H, W)], C, C, p, wp)
        _, _, vdec = OR.lif_scan_from_currents(tr["cur"])
        n_flip, margins, _ = first_flip_margins(planes_to_dense(spk, C), nchw_to_rows(tr["spk"]), nchw_to_rows(vdec))
        assert n_flip <= 2 + 1e-5 * tr["spk"][0].numel(), n_flip
        assert (margins <= TIE_MARGIN).all(), margins


@pytest.mark.parametrize("name", ["det_K9_T12", "det_small_T16", "det_K11_T8_R37"])
def test_linear_lif_on_period_planes_teacher_forced(gpu_device, monkeypatch, name):
    from snn_automotive_object_detection_amd import ops
    spec = FX.DET_SPECS[name]
    T, Hd = spec["T"], spec["Hd"]
    x, w6, w7, wc, wb = FX.det_inputs(spec)
    D = x[0].numel()
    p = _params(ops)
    _, _, tr = OR.det_head_forward(x, w6, w7, wc, wb, T, trace=True)
    monkeypatch.setenv("SNN_STAGE_PERIODS", "1")
    enc = ops.encode_rows(x.flatten(1).to(gpu_device), T, p)
    assert torch.equal(_reconstruct(enc).cpu(), dense_to_planes(tr["z"].numpy()))
    s6 = ops.spike_gemm_lif_bf16x3(enc, D, Hd, p, ops.pack_linear_bf16x3(w6.to(gpu_device)))
    _, _, vdec = OR.lif_scan_from_currents(tr["cur6"])
    n_flip, margins, _ = first_flip_margins(planes_to_dense(s6, Hd), tr["spk6"].numpy(), vdec.numpy())
    assert n_flip <= 2 + 1e-5 * tr["spk6"][0].numel(), n_flip
    assert (margins <= TIE_MARGIN).all(), margins


@pytest.mark.parametrize("T", [3, 8, 12, 16])
def test_rpn_head_period_mode_against_spike_plane_mode(gpu_device, monkeypatch, T):
    import snn_automotive_object_detection_amd as S
    torch.manual_seed(T)
    m = S.RPNHeadSNN(256, 3, T).to(gpu_device)
    with torch.no_grad():
        m.shared_conv.weight.mul_(3.0)
    m.spike_rates = True
    shapes = [(2, 48, 96), (2, 24, 43), (1, 12, 22), (2, 3, 5)]                     # (odd widths: BDD-like levels)
    feats = [torch.randn(n, 256, h, w, device=gpu_device) * 3.0 for n, h, w in shapes]

    def run():
        lg, bb, rates = m(feats)
        return [x.clone() for x in lg + bb], m.last_spike_counts.clone()
    per, cnt_p = run()
    monkeypatch.setenv("SNN_PERIOD_PLANES", "0")
    direct, cnt_d = run()
    assert int(cnt_p.sum()) > 0
    pos = off = 0
    L = len(shapes)
    for l in range(L):
        d = torch.maximum((per[l] - direct[l]).abs().amax(dim=1), (per[L + l] - direct[L + l]).abs().amax(dim=1))
        pos += d.numel(); off += int((d > 1e-4).sum())
    assert off <= flip_budget(pos, 256, T), (off, pos)
    assert int((cnt_p - cnt_d).abs().max()) <= 4 * flip_budget(pos, 256, T)


@pytest.mark.parametrize("R,T", [(300, 12), (64, 8), (500, 24), (37, 3)])
def test_det_head_period_mode_against_spike_plane_mode(gpu_device, monkeypatch, R, T):
    import snn_automotive_object_detection_amd as S
    from tests.test_gpu_roialign import _setup
    torch.manual_seed(R + T)
    m = S.FastRCNNPredictorSNNFull(64 * 49, 256, 9, T).to(gpu_device)
    with torch.no_grad():
        m.fc7.weight.mul_(3.0)
    x = torch.randn(R, 64, 7, 7, device=gpu_device) * 3.0
    pool, feats, boxes, shapes = _setup(gpu_device, R=max(R, 6), C=64, seed=R)
    flist, scales, rois, lvl = pool.assign(feats, boxes, shapes)

    def run():
        m.spike_rates = False
        c, b = m(x)
        c2, b2 = m.forward_roialign(flist, scales, rois, lvl)
        m.spike_rates = True
        m(x)
        return [c.clone(), b.clone(), c2.clone(), b2.clone()], [t.clone() for t in m.last_spike_counts]
    per, cnt_p = run()
    monkeypatch.setenv("SNN_PERIOD_PLANES", "0")
    direct, cnt_d = run()
    assert int(cnt_p[0].sum()) > 0
    for a, b, n in ((per[0], direct[0], R), (per[2], direct[2], rois.shape[0])):
        off = int(((a - b).abs().amax(dim=1) > 1e-4).sum())
        assert off <= flip_budget(n, 2 * 256, T, "det"), (off, n)
    assert int((cnt_p[0] != cnt_d[0]).sum()) <= flip_budget(R, 256, T, "det")


def test_generic_rest_potentials_and_the_other_precisions_take_spike_planes(gpu_device, monkeypatch):
    """period planes need the +0 start / reset state: any other parameters (C ABI only - the modules refuse them), SNN_ENC_GENERIC and
    the f32 / mxfp6 families multiply the spike planes; the knob then changes nothing, bit for bit"""
    import snn_automotive_object_detection_amd as S
    from snn_automotive_object_detection_amd import ops
    torch.manual_seed(1)
    m = S.RPNHeadSNN(128, 3, 8).to(gpu_device)
    with torch.no_grad():
        m.shared_conv.weight.mul_(4.0)
    feats = [torch.randn(2, 128, 9, 14, device=gpu_device) * 3.0]
    for prec, enc_generic in (("f32", None), ("mxfp6", None), ("bf16x3", "1")):
        m.precision = prec
        if enc_generic:
            monkeypatch.setenv("SNN_ENC_GENERIC", enc_generic)
        a = [t.clone() for t in m(feats)[0] + m(feats)[1]]
        monkeypatch.setenv("SNN_PERIOD_PLANES", "0")
        b = [t.clone() for t in m(feats)[0] + m(feats)[1]]
        monkeypatch.delenv("SNN_PERIOD_PLANES")
        assert all(torch.equal(x, y) for x, y in zip(a, b)), prec
    monkeypatch.delenv("SNN_ENC_GENERIC")
    p = _params(ops)
    p.v_reset = -0.05                                                               # a reset potential: the detector head through the C ABI wrappers
    d = S.FastRCNNPredictorSNNFull(32 * 49, 64, 5, 6).to(gpu_device)
    w6, w7, wh = d._packed(inner=0)
    x = torch.randn(40, 32, 7, 7, device=gpu_device) * 2
    a = ops.det_head_forward(x, 64, 5, 20, 6, p, w6, w7, wh)[:2]
    monkeypatch.setenv("SNN_PERIOD_PLANES", "0")
    b = ops.det_head_forward(x, 64, 5, 20, 6, p, w6, w7, wh)[:2]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_threshold_encoders_leave_the_heads_bit_identical(gpu_device, monkeypatch):
    """SNN_ENC_QUANT=0 (period planes from the recurrence) against the default (threshold table): same planes, hence the same bits out
    of both heads, the fused RoIAlign path included"""
    import snn_automotive_object_detection_amd as S
    from tests.test_gpu_roialign import _setup
    torch.manual_seed(2)
    r = S.RPNHeadSNN(128, 3, 8).to(gpu_device)
    d = S.FastRCNNPredictorSNNFull(32 * 49, 128, 9, 12).to(gpu_device)
    with torch.no_grad():
        r.shared_conv.weight.mul_(4.0)
    feats = [torch.randn(2, 128, 21, 30, device=gpu_device) * 3, torch.randn(2, 128, 5, 7, device=gpu_device) * 3]
    x = torch.randn(150, 32, 7, 7, device=gpu_device) * 3
    pool, fm, boxes, shapes = _setup(gpu_device, R=60, C=32, seed=4)
    flist, scales, rois, lvl = pool.assign(fm, boxes, shapes)

    def run():
        lg, bb = r(feats)
        c, b = d(x)
        c2, b2 = d.forward_roialign(flist, scales, rois, lvl)
        return [t.clone() for t in lg + bb] + [c.clone(), b.clone(), c2.clone(), b2.clone()]
    a = run()
    monkeypatch.setenv("SNN_ENC_QUANT", "0")
    b = run()
    assert all(torch.equal(p_, q_) for p_, q_ in zip(a, b))


@pytest.mark.parametrize("t_rpn,t_det", [(4, 8), (5, 9), (6, 10), (7, 11), (8, 12), (9, 13), (10, 14), (11, 15), (12, 16), (3, 17)])
def test_straight_line_epilogue_instances_equal_the_general_form(gpu_device, monkeypatch, t_rpn, t_det):
    """the T-in-tile LIF epilogue runs straight-line code instantiated per T (4 ... 16: conv window T - 1, fc6 window T - 2) and a
    guarded general form otherwise (SNN_EPI_GENERAL=1 forces it): the same operations in the same order - bit-identical heads over
    the paper's grid of time steps (metrics_for_different_timesteps.py:30-33) and one pair outside it"""
    import snn_automotive_object_detection_amd as S
    torch.manual_seed(t_rpn)
    r = S.RPNHeadSNN(128, 3, t_rpn).to(gpu_device)
    d = S.FastRCNNPredictorSNNFull(32 * 49, 128, 9, t_det).to(gpu_device)
    with torch.no_grad():
        r.shared_conv.weight.mul_(4.0)
    feats = [torch.randn(2, 128, 21, 30, device=gpu_device) * 3, torch.randn(2, 128, 5, 7, device=gpu_device) * 3]
    x = torch.randn(150, 32, 7, 7, device=gpu_device) * 3

    def run():
        lg, bb = r(feats)
        c, b = d(x)
        return [t.clone() for t in lg + bb] + [c.clone(), b.clone()]
    a = run()
    monkeypatch.setenv("SNN_EPI_GENERAL", "1")
    b = run()
    assert all(torch.equal(p_, q_) for p_, q_ in zip(a, b))
    assert any(float(t.abs().max()) > 0 for t in a)
