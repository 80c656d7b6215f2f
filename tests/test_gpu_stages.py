"""GPU parity, stage by stage and teacher-forced (SURVEY.md §7 risk 1): each HIP kernel is fed the
ORACLE's tensors of the previous stage and compared with the oracle's next stage.

  encoder kernels      bit-exact (element-wise fp32 arithmetic, same rounding sequence)
  LIF scan             bit-exact given identical input currents
  conv / GEMM currents <= 1e-5 abs (fp32 summation order differs from oneDNN's — cannot be bit-exact)
  conv+LIF spikes      identical except neurons whose oracle margin |v_dec - theta| is below the
                       current tolerance at their FIRST differing step (flip budget)
  LI heads             <= 1e-5 abs
All calls go through the C ABI (ctypes)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import fixtures as FX
from oracle import snn_oracle as OR
from tests._util import planes_to_dense, dense_to_planes, nchw_to_rows, flip_budget

pytestmark = pytest.mark.gpu

CUR_TOL = 1e-5


@pytest.fixture(scope="module")
def S():
    import snn_automotive_object_detection_amd as pkg
    from snn_automotive_object_detection_amd import ops
    return ops


def _params(ops, li_order="jump_first"):
    return ops.make_params(ops.LIFParameters(v_th=torch.tensor(0.25)),
                           ops.LIFParameters(alpha=100, v_th=torch.tensor(0.1)), 0.001, li_order)


@pytest.mark.parametrize("name", sorted(FX.RPN_SPECS))
def test_encoder_nchw_bit_exact(S, gpu_device, name):
    spec = FX.RPN_SPECS[name]
    feats, *_ = FX.rpn_inputs(spec)
    for f in feats:
        z = OR.encoder_spikes(f, spec["T"])                       # [T,N,C,H,W]
        planes = S.encode_nchw(f.to(gpu_device), spec["T"], _params(S))
        got = planes_to_dense(planes, f.shape[1])
        assert np.array_equal(got, nchw_to_rows(z))
        # padding channels (C..Cw*32) must be silent
        full = planes_to_dense(planes, planes.shape[2] * 32)
        assert full[:, :, f.shape[1]:].sum() == 0


@pytest.mark.parametrize("name", sorted(FX.DET_SPECS))
def test_encoder_rows_bit_exact(S, gpu_device, name):
    spec = FX.DET_SPECS[name]
    x = FX.det_inputs(spec)[0].flatten(1)
    z = OR.encoder_spikes(x, spec["T"])                           # [T,R,D]
    planes = S.encode_rows(x.to(gpu_device), spec["T"], _params(S))
    assert np.array_equal(planes_to_dense(planes, x.shape[1]), z.numpy())


@pytest.mark.parametrize("R,D,T", [(37, 12544, 12), (5, 64, 3), (300, 1024, 32), (1, 32, 1)])
def test_encoder_rows_word_per_lane_equals_ballot_form_and_oracle(S, gpu_device, monkeypatch, R, D, T):
    """D % 32 == 0 runs the word-per-lane kernel (k_encode_rows_w); SNN_ENC_ROWS=ballot forces the element-per-lane one"""
    g = torch.Generator().manual_seed(R + D)
    x = torch.randn(R, D, generator=g) * 1.5
    x[0, :7] = torch.tensor([0.25, 0.26, 0.3, 1.0, 2.5, 2.6, -1.0])[:min(7, D)]      # threshold neighbours
    xg = x.to(gpu_device)
    for generic in ("0", "1"):
        monkeypatch.setenv("SNN_ENC_GENERIC", generic)
        monkeypatch.delenv("SNN_ENC_ROWS", raising=False)
        a = S.encode_rows(xg, T, _params(S))
        monkeypatch.setenv("SNN_ENC_ROWS", "ballot")
        b = S.encode_rows(xg, T, _params(S))
        assert torch.equal(a, b)
    z = OR.encoder_spikes(x, T)
    assert np.array_equal(planes_to_dense(a, D), z.numpy())


def test_encoder_fast_path_equals_op_for_op(S, gpu_device, monkeypatch):
    """v_leak = v_reset = 0 takes a 5-operation encoder step; SNN_ENC_GENERIC=1 forces the op-for-op kernels"""
    g = torch.Generator().manual_seed(5)
    f = (torch.randn(2, 70, 13, 17, generator=g) * 2.0).to(gpu_device)
    x = (torch.randn(37, 300, generator=g) * 2.0).to(gpu_device)
    # values that sit exactly on / next to the threshold after one step: v1 = 0.1 * x
    x[0, :4] = torch.tensor([2.5, 2.4999998, 2.5000002, 0.0])
    fast = S.encode_nchw(f, 16, _params(S)), S.encode_rows(x, 16, _params(S))
    monkeypatch.setenv("SNN_ENC_GENERIC", "1")
    slow = S.encode_nchw(f, 16, _params(S)), S.encode_rows(x, 16, _params(S))
    assert torch.equal(fast[0], slow[0]) and torch.equal(fast[1], slow[1])
    assert fast[0].ne(0).any() and fast[1].ne(0).any()


def _first_flip_margins(spk_got, spk_exp, vdec_exp, theta=0.1):
    """for every neuron whose spike train differs: oracle margin at the first differing step"""
    diff = spk_got != spk_exp                                     # [T, ...]
    any_diff = diff.any(axis=0)
    first = diff.argmax(axis=0)
    idx = np.nonzero(any_diff)
    margins = np.abs(vdec_exp[(first[idx],) + idx] - theta)
    return int(any_diff.sum()), margins


@pytest.mark.parametrize("name", sorted(FX.RPN_SPECS))
def test_conv3x3_lif_teacher_forced(S, gpu_device, name):
    spec = FX.RPN_SPECS[name]
    T, C = spec["T"], spec["C"]
    feats, w_s, w_c, w_b = FX.rpn_inputs(spec)
    wp = S.pack_conv3x3(w_s.to(gpu_device))
    _, _, traces = OR.rpn_head_forward(feats, w_s, w_c, w_b, T, trace=True)
    for f, tr in zip(feats, traces):
        N, _, H, W = f.shape
        enc = dense_to_planes(nchw_to_rows(tr["z"])).to(gpu_device)          # oracle's encoder spikes
        spk, counts, cur = S.conv3x3_lif(enc, N, C, C, H, W, _params(S), wp, want_counts=True, want_currents=True)
        # (i) input currents of every step
        cur_exp = nchw_to_rows(tr["cur"])
        cur_got = cur.cpu().numpy()[:, :, :C]
        assert np.abs(cur_got - cur_exp).max() <= CUR_TOL
        # (ii) spikes, with the flip budget
        spk_exp = nchw_to_rows(tr["spk"])
        spk_got = planes_to_dense(spk, C)
        _, _, vdec = OR.lif_scan_from_currents(tr["cur"])
        n_flip, margins = _first_flip_margins(spk_got, spk_exp, nchw_to_rows(vdec))
        assert n_flip <= 2 + 1e-5 * spk_exp[0].size, "too many flipped neurons: %d" % n_flip
        assert (margins <= 2 * CUR_TOL).all(), margins
        # (iii) spike counts == popcount of what was written
        got_counts = spk_got.reshape(T, N, H * W, C).sum(axis=(0, 2, 3))
        assert np.array_equal(counts.cpu().numpy(), got_counts.astype(np.int64))


@pytest.mark.parametrize("name", sorted(FX.DET_SPECS))
def test_det_stages_teacher_forced(S, gpu_device, name):
    spec = FX.DET_SPECS[name]
    T, Hd, K = spec["T"], spec["Hd"], spec["K"]
    x, w6, w7, wc, wb = FX.det_inputs(spec)
    R, D = x.shape[0], x[0].numel()
    _, _, tr = OR.det_head_forward(x, w6, w7, wc, wb, T, trace=True)
    p = _params(S)
    # fc6 currents from the oracle's encoder spikes, all T steps in one GEMM (rows t*R + r)
    zp = dense_to_planes(tr["z"].numpy()).to(gpu_device)
    cur6 = S.spike_gemm(zp.view(T * R, -1), D, Hd, S.pack_linear(w6.to(gpu_device)))
    cur6 = cur6.view(T, R, -1)[:, :, :Hd].cpu()
    assert (cur6 - tr["cur6"]).abs().max() <= CUR_TOL
    # LIF scan on the ORACLE's currents is bit-exact (element-wise arithmetic only)
    spk6, c6 = S.lif_scan(tr["cur6"].to(gpu_device), Hd, p, want_counts=True)
    assert np.array_equal(planes_to_dense(spk6, Hd), tr["spk6"].numpy())
    assert np.array_equal(c6.cpu().numpy(), tr["spk6"].sum(dim=(0, 2)).numpy().astype(np.int32))
    # fc7 from the oracle's spk6
    s6p = dense_to_planes(tr["spk6"].numpy()).to(gpu_device)
    cur7 = S.spike_gemm(s6p.view(T * R, -1), Hd, Hd, S.pack_linear(w7.to(gpu_device))).view(T, R, -1)[:, :, :Hd].cpu()
    assert (cur7 - tr["cur7"]).abs().max() <= CUR_TOL
    spk7 = S.lif_scan(tr["cur7"].to(gpu_device), Hd, p)
    assert np.array_equal(planes_to_dense(spk7, Hd), tr["spk7"].numpy())
    # LI heads from the oracle's spk7: last membranes and their sums over t
    s7p = dense_to_planes(tr["spk7"].numpy()).to(gpu_device)
    wh = S.pack_heads(wc.to(gpu_device), wb.to(gpu_device))
    o_c, o_b, s_c, s_b = S.li_heads(s7p, Hd, wh, wc.shape[0], wb.shape[0], p, want_sums=True)
    assert (o_c.cpu() - tr["mem_cls"][-1]).abs().max() <= CUR_TOL
    assert (o_b.cpu() - tr["mem_bbox"][-1]).abs().max() <= CUR_TOL
    assert (s_c.cpu() - tr["mem_cls"].sum(0)).abs().max() <= T * CUR_TOL
    assert (s_b.cpu() - tr["mem_bbox"].sum(0)).abs().max() <= T * CUR_TOL


@pytest.mark.parametrize("li_order", ["jump_first", "voltage_first"])
def test_li_heads_both_orders(S, gpu_device, li_order):
    spec = FX.RPN_SPECS["rpn_c64_A5_T8"]
    T, C, A = spec["T"], spec["C"], spec["A"]
    feats, w_s, w_c, w_b = FX.rpn_inputs(spec)
    _, _, traces = OR.rpn_head_forward(feats, w_s, w_c, w_b, T, trace=True)
    spk = traces[0]["spk"]
    exp_o, _ = OR.li_last_from_spikes(spk, w_c, li_order, conv=True)
    exp_b, _ = OR.li_last_from_spikes(spk, w_b, li_order, conv=True)
    planes = dense_to_planes(nchw_to_rows(spk)).to(gpu_device)
    wh = S.pack_heads(w_c.to(gpu_device), w_b.to(gpu_device))
    o_a, o_b = S.li_heads(planes, C, wh, A, 4 * A, _params(S, li_order))
    N, _, H, W = exp_o.shape
    assert (o_a.cpu().view(N, H, W, A).permute(0, 3, 1, 2) - exp_o).abs().max() <= CUR_TOL
    assert (o_b.cpu().view(N, H, W, 4 * A).permute(0, 3, 1, 2) - exp_b).abs().max() <= CUR_TOL


@pytest.mark.parametrize("T,M,K,NA,NB", [(12, 300, 1024, 9, 36), (16, 45, 160, 2, 8), (12, 130, 1024, 11, 44), (5, 17, 2048, 3, 12),
                                          (24, 70, 1024, 9, 36), (20, 33, 512, 11, 44),      # these two: more than one time group
                                          (12, 50, 256, 24, 96), (10, 21, 1024, 91, 364)])   # more than 64 outputs: blocks of 64 columns
@pytest.mark.parametrize("li_order", ["jump_first", "voltage_first"])
def test_li_heads_kernel_forms_agree_with_fp64(S, gpu_device, monkeypatch, li_order, T, M, K, NA, NB):
    """the three kernels behind snn_li_heads (fp32 VALU, matrix cores with W resident in LDS, matrix cores with the reduction
    split over the waves of a 16-row work-group) against the LI recursion in fp64 on the same spikes"""
    g = torch.Generator().manual_seed(T * M + K)
    spk = torch.rand(T, M, K, generator=g) < 0.1
    wa = torch.randn(NA, K, generator=g) / K ** 0.5
    wb = torch.randn(NB, K, generator=g) / K ** 0.5
    a, b = float(torch.tensor(0.001) * torch.tensor(100.0)), float(torch.tensor(0.001) * torch.tensor(200.0))
    cur = torch.einsum("tmk,nk->tmn", spk.double(), torch.cat([wa, wb]).double())
    v = torch.zeros(M, NA + NB, dtype=torch.float64)
    i = torch.zeros_like(v)
    vsum = torch.zeros_like(v)
    for t in range(T):
        if li_order == "jump_first":
            i = i + cur[t]; v = v + a * (i - v); i = i - b * i
        else:
            v = v + a * (i - v); i = i - b * i + cur[t]
        vsum = vsum + v
    planes = dense_to_planes(spk.numpy()).to(gpu_device)
    wh = S.pack_heads(wa.to(gpu_device), wb.to(gpu_device))
    outs = {}
    for form in ("valu", "ksplit", "mfma"):
        monkeypatch.setenv("SNN_LI_HEADS", form)      # a form that does not apply to the shape falls through to the VALU kernel
        o_a, o_b, s_a, s_b = S.li_heads(planes, K, wh, NA, NB, _params(S, li_order), want_sums=True)
        got = torch.cat([o_a, o_b], dim=1).double().cpu()
        gsum = torch.cat([s_a, s_b], dim=1).double().cpu()
        assert float((got - v).abs().max()) <= CUR_TOL, form
        assert float((gsum - vsum).abs().max()) <= T * CUR_TOL, form
        outs[form] = got
    assert float((outs["valu"] - outs["ksplit"]).abs().max()) <= 2 * CUR_TOL


@pytest.mark.parametrize("K,NA,NB", [(256, 3, 12), (256, 9, 36), (128, 5, 20), (1024, 9, 36), (512, 2, 8)])
def test_li_heads_every_number_of_steps(S, gpu_device, monkeypatch, K, NA, NB):
    """T = 1 .. 26 on every kernel form: the time steps run in groups (8; 16 / 12 on the reduction-split kernel), so every length of
    the last group is exercised.  (Round 4 found the matrix-core kernel wrong for last groups of exactly two steps - T = 2, 10, 18, 26 -
    at C = 256: rows 3 mod 4 of every 16-row tile; no other test ran those T.)"""
    M = 83
    a, b = float(torch.tensor(0.001) * torch.tensor(100.0)), float(torch.tensor(0.001) * torch.tensor(200.0))
    g = torch.Generator().manual_seed(K + NA)
    wa = torch.randn(NA, K, generator=g) / K ** 0.5
    wb = torch.randn(NB, K, generator=g) / K ** 0.5
    wh = S.pack_heads(wa.to(gpu_device), wb.to(gpu_device))
    spk_all = torch.rand(26, M, K, generator=g) < 0.1
    cur_all = torch.einsum("tmk,nk->tmn", spk_all.double(), torch.cat([wa, wb]).double())
    for T in range(1, 27):
        v = torch.zeros(M, NA + NB, dtype=torch.float64)
        i = torch.zeros_like(v)
        vsum = torch.zeros_like(v)
        for t in range(T):
            i = i + cur_all[t]; v = v + a * (i - v); i = i - b * i
            vsum = vsum + v
        planes = dense_to_planes(spk_all[:T].numpy()).to(gpu_device)
        for form in ("valu", "ksplit", "mfma"):
            monkeypatch.setenv("SNN_LI_HEADS", form)
            o_a, o_b, s_a, s_b = S.li_heads(planes, K, wh, NA, NB, _params(S, "jump_first"), want_sums=True)
            got = torch.cat([o_a, o_b], dim=1).double().cpu()
            gsum = torch.cat([s_a, s_b], dim=1).double().cpu()
            assert float((got - v).abs().max()) <= CUR_TOL, (form, T)
            assert float((gsum - vsum).abs().max()) <= T * CUR_TOL, (form, T)


# ---------------------------------------------------------------------------------------------
# exact bf16x3 contractions (bf16 matrix cores): same teacher-forced bars as the fp32 MFMA kernels
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", sorted(FX.DET_SPECS))
def test_spike_gemm_bf16x3_teacher_forced(S, gpu_device, name):
    spec = FX.DET_SPECS[name]
    T, Hd = spec["T"], spec["Hd"]
    x, w6, w7, wc, wb = FX.det_inputs(spec)
    R, D = x.shape[0], x[0].numel()
    _, _, tr = OR.det_head_forward(x, w6, w7, wc, wb, T, trace=True)
    zp = dense_to_planes(tr["z"].numpy()).to(gpu_device)
    cur6 = S.spike_gemm_bf16x3(zp.view(T * R, -1), D, Hd, S.pack_linear_bf16x3(w6.to(gpu_device)))
    assert (cur6.view(T, R, -1)[:, :, :Hd].cpu() - tr["cur6"]).abs().max() <= CUR_TOL
    s6p = dense_to_planes(tr["spk6"].numpy()).to(gpu_device)
    cur7 = S.spike_gemm_bf16x3(s6p.view(T * R, -1), Hd, Hd, S.pack_linear_bf16x3(w7.to(gpu_device)))
    assert (cur7.view(T, R, -1)[:, :, :Hd].cpu() - tr["cur7"]).abs().max() <= CUR_TOL
    # the two kernel families agree far below the tolerance
    cur6_f32 = S.spike_gemm(zp.view(T * R, -1), D, Hd, S.pack_linear(w6.to(gpu_device)))
    assert (cur6 - cur6_f32).abs().max() <= 5e-6


@pytest.mark.parametrize("name", sorted(FX.DET_SPECS))
def test_spike_gemm_lif_bf16x3_teacher_forced(S, gpu_device, name):
    """fc + LIF fused over T inside the row tile: spikes of fc6 / fc7 from the oracle's input spikes"""
    spec = FX.DET_SPECS[name]
    T, Hd = spec["T"], spec["Hd"]
    x, w6, w7, wc, wb = FX.det_inputs(spec)
    R, D = x.shape[0], x[0].numel()
    _, _, tr = OR.det_head_forward(x, w6, w7, wc, wb, T, trace=True)
    p = _params(S)
    zp = dense_to_planes(tr["z"].numpy()).to(gpu_device)
    s6 = S.spike_gemm_lif_bf16x3(zp, D, Hd, p, S.pack_linear_bf16x3(w6.to(gpu_device)))
    got6 = planes_to_dense(s6, Hd)
    assert (got6 != tr["spk6"].numpy()).any(axis=(0, 2)).sum() <= flip_budget(R, Hd, T)
    # identical to the un-fused pair on the same operands (same accumulation order, same LIF arithmetic)
    cur6 = S.spike_gemm_bf16x3(zp.view(T * R, -1), D, Hd, S.pack_linear_bf16x3(w6.to(gpu_device)))
    ref6 = S.lif_scan(cur6.view(T, R, -1), Hd, p)
    assert torch.equal(ref6[..., :s6.shape[-1]], s6)
    s6p = dense_to_planes(tr["spk6"].numpy()).to(gpu_device)
    s7 = S.spike_gemm_lif_bf16x3(s6p, Hd, Hd, p, S.pack_linear_bf16x3(w7.to(gpu_device)))
    assert (planes_to_dense(s7, Hd) != tr["spk7"].numpy()).any(axis=(0, 2)).sum() <= flip_budget(R, Hd, T)


@pytest.mark.parametrize("name", sorted(FX.RPN_SPECS))
def test_spike_conv3x3_bf16x3_teacher_forced(S, gpu_device, name):
    spec = FX.RPN_SPECS[name]
    T, C = spec["T"], spec["C"]
    feats, w_s, w_c, w_b = FX.rpn_inputs(spec)
    _, _, traces = OR.rpn_head_forward(feats, w_s, w_c, w_b, T, trace=True)
    wp = S.pack_conv3x3_bf16x3(w_s.to(gpu_device))
    # all levels in ONE launch: planes concatenated along the position axis
    enc = torch.cat([dense_to_planes(nchw_to_rows(tr["z"])) for tr in traces], dim=1).to(gpu_device)
    shapes = [(f.shape[0], f.shape[2], f.shape[3]) for f in feats]
    cur = S.spike_conv3x3_bf16x3(enc, shapes, C, C, wp).cpu().numpy()
    pos = 0
    for f, tr in zip(feats, traces):
        n = f.shape[0] * f.shape[2] * f.shape[3]
        exp = nchw_to_rows(tr["cur"])
        assert np.abs(cur[:, pos:pos + n, :C] - exp).max() <= CUR_TOL
        pos += n
    # LIF scan on these currents reproduces the oracle's spikes up to threshold ties
    p = _params(S)
    spk = S.lif_scan(torch.from_numpy(cur).to(gpu_device), C, p)
    got = planes_to_dense(spk, C)
    exp_spk = np.concatenate([nchw_to_rows(tr["spk"]) for tr in traces], axis=1)
    flipped = (got != exp_spk).any(axis=(0, 2)).sum()
    assert flipped <= 2 + 1e-3 * got.shape[1]
    # the fused conv + LIF kernel (state in registers over the T loop) gives the same spike planes as the
    # un-fused pair bit for bit: same MFMA order, same element-wise arithmetic
    fused = S.conv3x3_lif_bf16x3(enc, shapes, C, C, p, wp)
    assert torch.equal(fused, spk)


@pytest.mark.parametrize("T", [4, 12])
def test_bf16x3_kernel_variants_bit_identical(S, gpu_device, monkeypatch, T):
    """The fallbacks / alternative tilings of k_gemm_bf16x3 give the same bits as the default launch: register-resident
    LIF (SNN_BF16X3_LIF=reg, the any-T fallback) vs T-in-tile fusion, the 4x2 / 8x1 wave grids (SNN_BF16X3_WN) and the
    smaller work-group tiles (SNN_BF16X3_MT) - same products, same accumulation order per output, same LIF arithmetic."""
    g = torch.Generator().manual_seed(11 + T)
    C = 96
    shapes = [(2, 9, 14), (2, 5, 7), (1, 3, 3)]
    P = sum(n * h * w for n, h, w in shapes)
    p = _params(S)
    enc = torch.randint(-2**31, 2**31 - 1, (T, P, 3), generator=g, dtype=torch.int64).to(torch.int32)
    enc = (enc & torch.randint(-2**31, 2**31 - 1, (T, P, 3), generator=g, dtype=torch.int64).to(torch.int32)).to(gpu_device)
    w = (torch.randn(C, C, 3, 3, generator=g) * 0.05).to(gpu_device)
    wp = S.pack_conv3x3_bf16x3(w)
    base = S.conv3x3_lif_bf16x3(enc, shapes, C, C, p, wp)
    assert base.ne(0).any()
    for wn in ("1", "2"):
        monkeypatch.setenv("SNN_BF16X3_WN", wn)
        for mt in ("2", "3", "4"):
            monkeypatch.setenv("SNN_BF16X3_MT", mt)
            assert torch.equal(S.conv3x3_lif_bf16x3(enc, shapes, C, C, p, wp), base), "WN=%s MT=%s" % (wn, mt)
    monkeypatch.setenv("SNN_BF16X3_WN", "2")
    monkeypatch.setenv("SNN_BF16X3_MT", "8")                      # fat waves: 4 waves of 128 x 64 on the same 256 x 128 tile
    assert torch.equal(S.conv3x3_lif_bf16x3(enc, shapes, C, C, p, wp), base), "fat waves"
    monkeypatch.delenv("SNN_BF16X3_MT")
    monkeypatch.delenv("SNN_BF16X3_WN")
    monkeypatch.setenv("SNN_BF16X3_LIF", "reg")
    assert torch.equal(S.conv3x3_lif_bf16x3(enc, shapes, C, C, p, wp), base)
    monkeypatch.delenv("SNN_BF16X3_LIF")
    # linear layer + LIF: tile variants, and the un-fused pair
    R, K, N = 45, 200, 70
    a = torch.randint(-2**31, 2**31 - 1, (T, R, 7), generator=g, dtype=torch.int64).to(torch.int32).to(gpu_device)
    a[..., 6] &= 0xFF                                             # K = 200: bits past K are zero in real planes
    wl = (torch.randn(N, K, generator=g) * 0.1).to(gpu_device)
    wlp = S.pack_linear_bf16x3(wl)
    base = S.spike_gemm_lif_bf16x3(a, K, N, p, wlp)
    for wn in ("1", "2"):
        monkeypatch.setenv("SNN_BF16X3_WN", wn)
        for mt in ("2", "3", "4"):
            monkeypatch.setenv("SNN_BF16X3_MT", mt)
            assert torch.equal(S.spike_gemm_lif_bf16x3(a, K, N, p, wlp), base), "WN=%s MT=%s" % (wn, mt)
            cur = S.spike_gemm_bf16x3(a.view(T * R, -1), K, N, wlp)
            assert torch.equal(S.lif_scan(cur.view(T, R, -1), N, p)[..., :base.shape[-1]], base)
    monkeypatch.setenv("SNN_BF16X3_WN", "2")
    monkeypatch.setenv("SNN_BF16X3_MT", "8")
    assert torch.equal(S.spike_gemm_lif_bf16x3(a, K, N, p, wlp), base), "fat waves"
    cur = S.spike_gemm_bf16x3(a.view(T * R, -1), K, N, wlp)
    assert torch.equal(S.lif_scan(cur.view(T, R, -1), N, p)[..., :base.shape[-1]], base)


@pytest.mark.parametrize("C_in,C_out,T,shapes", [
    (40, 200, 8, [(1, 7, 9), (2, 3, 4)]),          # padded input word, two column blocks, the second one partial
    (256, 32, 5, [(1, 16, 16)]),                   # one 32-channel output word: half of a 64-column epilogue half
    (33, 129, 16, [(3, 5, 5), (1, 1, 1), (1, 2, 9)]),
    (64, 64, 24, [(1, 11, 13)]),                   # T = 24: 10 positions per 256-row tile, 16 rows idle
    (96, 320, 3, [(2, 6, 6)]),                     # three column blocks
])
def test_conv3x3_lif_bf16x3_odd_shapes_equal_unfused_pair(S, gpu_device, C_in, C_out, T, shapes):
    """T-in-tile fused conv + LIF == un-fused conv GEMM followed by the LIF scan, bit for bit, on shapes that exercise
    channel padding, partial column blocks, 1x1 levels and T values that do not divide the row tile"""
    g = torch.Generator().manual_seed(C_in * 1000 + C_out + T)
    P = sum(n * h * w for n, h, w in shapes)
    Cw = (C_in + 31) // 32
    bits = torch.rand(T, P, Cw * 32, generator=g) < 0.2
    bits[..., C_in:] = False
    enc = dense_to_planes(bits.numpy().astype(np.float32)).to(gpu_device)
    w = (torch.randn(C_out, C_in, 3, 3, generator=g) * 0.08).to(gpu_device)
    wp = S.pack_conv3x3_bf16x3(w)
    p = _params(S)
    fused = S.conv3x3_lif_bf16x3(enc, shapes, C_in, C_out, p, wp)
    cur = S.spike_conv3x3_bf16x3(enc, shapes, C_in, C_out, wp)
    ref = S.lif_scan(cur, C_out, p)
    assert fused.shape == ref.shape and torch.equal(fused, ref)
    assert fused.ne(0).any()
    # and the currents are the convolution (fp64 reference on the host)
    x = bits[..., :C_in].double()
    pos = 0
    for n, h, wd in shapes:
        xi = x[:, pos:pos + n * h * wd].reshape(T * n, h, wd, C_in).permute(0, 3, 1, 2)
        exp = F.conv2d(xi, w.double().cpu(), padding=1).permute(0, 2, 3, 1).reshape(T, n * h * wd, C_out)
        got = cur[:, pos:pos + n * h * wd, :C_out].double().cpu()
        assert (got - exp).abs().max() <= 1e-5
        pos += n * h * wd


def _popcount_rows(planes: torch.Tensor) -> torch.Tensor:
    """int32 [T, M, W] -> int64 [M]: set bits per row over all planes and words"""
    lut = torch.tensor([bin(i).count("1") for i in range(256)], dtype=torch.int64, device=planes.device)
    return lut[planes.contiguous().view(torch.uint8).to(torch.int64)].view(planes.shape[0], planes.shape[1], -1).sum(dim=(0, 2))


@pytest.mark.parametrize("precision,C", [("bf16x3", 96), ("mxfp6", 128), ("f32", 96)])
def test_fused_spike_counts_equal_popcount_of_the_planes(gpu_device, monkeypatch, precision, C):
    """the spike counts that the LIF epilogues add up with ballot / popcount / integer atomics (round 2) against a popcount
    of the spike planes the same kernels write, per (level, image) and per RoI, over the tile shapes and wave grids - incl.
    levels of 1 and 6 positions per image, where one tile spans many (level, image) slots"""
    import snn_automotive_object_detection_amd as pkg
    from snn_automotive_object_detection_amd import ops
    g = torch.Generator().manual_seed(31)
    T = 8
    shapes = [(3, 9, 14), (3, 2, 3), (3, 1, 1), (3, 5, 7)]
    feats = [(torch.randn((n, C, h, w), generator=g) * 2).to(gpu_device) for n, h, w in shapes]
    m = pkg.RPNHeadSNN(C, 3, T).to(gpu_device)
    with torch.no_grad():
        m.shared_conv.weight.mul_(6.0)                         # make the shared LIF fire
    m.precision = precision
    m.spike_rates = True
    p = m._params()
    variants = [("2", "4")] if precision != "bf16x3" else [("2", "4"), ("2", "3"), ("2", "2"), ("1", "4"), ("1", "2")]
    for wn, mt in variants:
        monkeypatch.setenv("SNN_BF16X3_WN", wn)
        monkeypatch.setenv("SNN_BF16X3_MT", mt)
        m(feats)
        counts = m.last_spike_counts.cpu()
        # the same planes through the stage ops, counted on the side
        encs = torch.cat([ops.encode_nchw(f, T, p) for f in feats], dim=1)
        if precision == "f32":
            rows = torch.cat([_popcount_rows(ops.conv3x3_lif(ops.encode_nchw(f, T, p), n, C, C, h, w, p, ops.pack_conv3x3(m.shared_conv.weight)))
                              for f, (n, h, w) in zip(feats, shapes)])
        elif precision == "mxfp6":
            rows = _popcount_rows(ops.conv3x3_lif_mx(ops.pad_planes(encs, shapes), shapes, C, C, p, ops.pack_conv3x3_mx(m.shared_conv.weight)))
        else:
            rows = _popcount_rows(ops.conv3x3_lif_bf16x3(encs, shapes, C, C, p, ops.pack_conv3x3_bf16x3(m.shared_conv.weight)))
        pos = 0
        for l, (n, h, w) in enumerate(shapes):
            per_img = rows[pos:pos + n * h * w].view(n, -1).sum(1).cpu()
            assert torch.equal(counts[l, :n], per_img), (precision, wn, mt, l, counts[l, :n], per_img)
            assert int(per_img.sum()) > 0 or h * w == 1
            pos += n * h * w
    monkeypatch.delenv("SNN_BF16X3_WN")
    monkeypatch.delenv("SNN_BF16X3_MT")
    # detector: per-RoI counts of lif6 / lif7
    D = C * 49 if precision != "mxfp6" else 128 * 49
    Hd = 128
    d = pkg.FastRCNNPredictorSNNFull(D, Hd, 5, 12).to(gpu_device)
    with torch.no_grad():
        d.fc6.weight.mul_(4.0); d.fc7.weight.mul_(6.0)
    d.precision = precision
    d.spike_rates = True
    x = (torch.randn((37, D // 49, 7, 7), generator=g) * 2).to(gpu_device)
    d(x)
    c6, c7 = [c.cpu().to(torch.int64) for c in d.last_spike_counts]
    pd = d._params()
    enc = ops.encode_rows(x.flatten(1), 12, pd)
    if precision == "f32":
        s6 = ops.lif_scan(ops.spike_gemm(enc.view(12 * 37, -1), D, Hd, ops.pack_linear(d.fc6.weight)).view(12, 37, -1), Hd, pd)
        s7 = ops.lif_scan(ops.spike_gemm(s6.view(12 * 37, -1), Hd, Hd, ops.pack_linear(d.fc7.weight)).view(12, 37, -1), Hd, pd)
    elif precision == "mxfp6":
        s6 = ops.spike_gemm_lif_mx(enc, D, Hd, pd, ops.pack_linear_mx(d.fc6.weight))
        s7 = ops.spike_gemm_lif_mx(s6, Hd, Hd, pd, ops.pack_linear_mx(d.fc7.weight))
    else:
        s6 = ops.spike_gemm_lif_bf16x3(enc, D, Hd, pd, ops.pack_linear_bf16x3(d.fc6.weight))
        s7 = ops.spike_gemm_lif_bf16x3(s6, Hd, Hd, pd, ops.pack_linear_bf16x3(d.fc7.weight))
    assert torch.equal(c6, _popcount_rows(s6).cpu()) and torch.equal(c7, _popcount_rows(s7).cpu())
    assert int(c6.sum()) > 0 and int(c7.sum()) > 0
