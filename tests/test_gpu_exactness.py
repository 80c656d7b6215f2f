"""The two parity claims that used to rest on prose (VERDICT r3, P-a / P-b), pinned on the GPU:

 (a) "every fp32 weight is exactly hi + mid + lo with three bf16 values": the packed planes are read back and re-added BITWISE; a
     spike GEMM with ONE bit set per row must return the weight rows BITWISE (that also answers what the bf16 matrix cores do with
     subnormal operands); and where the split cannot be exact (bits below 2^-133, values next to FLT_MAX, NaN / infinity) the pack
     call refuses and the modules run "f32_strict" with a RuntimeWarning.
 (b) non-finite features: the reference's encoder resets arithmetically, v - z * (v - v_reset) (norse lif_current_encoder,
     /root/reference/rpn.py:101, faster_rcnn.py:494), so a +inf feature spikes ONCE and is NaN afterwards; the product's encoders
     reset by selection and make it a period-1 neuron.  -inf, NaN, huge finite values and -0.0 behave as in the reference.  This is
     the documented divergence ("Non-finite features"); SNN_ENC_GENERIC=1 runs the reference's operations one by
     one and reproduces its trains.  The oracle side of the same statement: tests/test_oracle_kat.py."""
import warnings

import numpy as np
import pytest
import torch

from oracle import snn_oracle as OR
from tests._util import planes_to_dense, dense_to_planes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from snn_automotive_object_detection_amd import ops
    return ops


def _bits(a: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _weights_exact_class(n: int, seed: int) -> np.ndarray:
    """fp32 bit patterns the split must carry exactly: random normal-range values over ~all exponents, +-0, powers of two, values with
    all 24 mantissa bits set, magnitudes down to 2^-109 with full mantissas and below that as long as no bit lies under 2^-133"""
    rng = np.random.default_rng(seed)
    n_extra = 14 + 3 * 64
    assert n > n_extra
    n_rand = n - n_extra
    exp = rng.integers(127 - 109, 127 + 126, size=n_rand).astype(np.uint32)      # unbiased -109 .. +125
    man = rng.integers(0, 1 << 23, size=n_rand).astype(np.uint32)
    sign = rng.integers(0, 2, size=n_rand).astype(np.uint32)
    w = ((sign << 31) | (exp << 23) | man).view(np.float32)
    special = np.array([0.0, -0.0, 1.0, -1.0, 0.01, -0.01, 3.0e38, -3.0e38, 2.0 ** -100, 2.0 ** -109, 1.5 * 2.0 ** -109,
                        np.float32(2.0 ** -110) * np.float32(1.9999999), 2.0 ** 127, np.float32(1.9921875) * np.float32(2.0 ** 127)],
                       dtype=np.float32)
    allset = ((rng.integers(18, 250, size=64).astype(np.uint32) << 23) | np.uint32(0x7fffff)).view(np.float32)
    # tiny values with few significant bits: k * 2^-133 (k < 2^7: one subnormal bf16 plane), and m * 2^-126 * 2^-7 style values
    tiny = (rng.integers(1, 128, size=64).astype(np.float64) * 2.0 ** -133).astype(np.float32)
    tiny2 = (rng.integers(1, 1 << 16, size=64).astype(np.float64) * 2.0 ** -133).astype(np.float32) * np.float32(2.0 ** 8)   # multiples of 2^-125
    out = np.concatenate([w, special, allset, tiny, tiny2]).astype(np.float32)
    assert out.size == n
    return out[rng.permutation(n)]                                               # exactly n values, the special ones spread over the tensor


def _weights_inexact_class() -> np.ndarray:
    """finite values the three planes cannot carry: bits below 2^-133, and the top of the range where hi rounds up to infinity"""
    sub = np.array([1, 3, 0x7fff, 0x12345, 0x7fffff], dtype=np.uint32).view(np.float32)            # fp32 subnormals with low bits
    tiny_normal = np.array([0x00800001, 0x00ffffff, 0x01000001, 0x04800001], dtype=np.uint32).view(np.float32)   # ulp < 2^-133
    top = np.array([0x7f7fffff, 0x7f7f8000, 0xff7fffff], dtype=np.uint32).view(np.float32)         # hi = rn_bf16(w) = inf
    return np.concatenate([sub, tiny_normal, top])


def _unpack_planes(packed: torch.Tensor, N: int, K: int):
    """uint16 [3][Kc][Np][32] -> three float32 [N][K] arrays (hi, mid, lo)"""
    Kc, Np = (K + 31) // 32, (N + 31) // 32 * 32
    a = packed.cpu().numpy().view(np.uint16).reshape(3, Kc, Np, 32)
    f = (a.astype(np.uint32) << 16).view(np.float32)                              # bf16 -> fp32, exact
    f = f.transpose(0, 2, 1, 3).reshape(3, Np, Kc * 32)[:, :N, :K]
    return f[0], f[1], f[2]


def test_pack_bf16x3_is_exact(ops, gpu_device):
    """hi + mid + lo == w, bit for bit, for every weight of the exact class - through BOTH pack entry points - and the planes really are
    bf16 roundings (|mid| <= ulp_bf16(hi) / 2 ...: implied by exactness with three 8-bit significands)"""
    N, K = 96, 625
    w = _weights_exact_class(N * K, 1).reshape(N, K)
    wt = torch.from_numpy(w.copy()).to(gpu_device)
    assert ops.bf16x3_split_status(wt)[:2] == (0, 0)
    hi, mid, lo = _unpack_planes(ops.pack_linear_bf16x3(wt), N, K)
    back = (lo + mid) + hi                                                        # the kernels' accumulation order: small terms first
    nz = w != 0
    assert np.array_equal(_bits(back)[nz], _bits(w)[nz])
    assert np.all(back[~nz] == 0)                                                 # +-0: equal as values (the sum of -0 + 0 + 0 is +0)
    assert np.array_equal(_bits((hi + mid) + lo)[nz], _bits(w)[nz])               # ... and in the other order
    # 3x3 conv packing: k = tap * Cp + ci
    Co, Ci = 40, 24
    wc = _weights_exact_class(Co * Ci * 9, 2).reshape(Co, Ci, 3, 3)
    pk = ops.pack_conv3x3_bf16x3(torch.from_numpy(wc.copy()).to(gpu_device))
    Cp = 32
    h3, m3, l3 = _unpack_planes(pk, Co, 9 * Cp)
    back = ((l3 + m3) + h3).reshape(Co, 9, Cp)[:, :, :Ci].transpose(0, 2, 1).reshape(Co, Ci, 3, 3)
    nz = wc != 0
    assert np.array_equal(_bits(back)[nz], _bits(wc)[nz]) and np.all(back[~nz] == 0)
    pad = ((l3 + m3) + h3).reshape(Co, 9, Cp)[:, :, Ci:]
    assert np.all(pad == 0)


def test_split_check_flags_what_cannot_be_exact(ops, gpu_device):
    bad = _weights_inexact_class()
    for v in bad:
        st = ops.bf16x3_split_status(torch.tensor([[1.0, float(v), 0.5]], device=gpu_device))
        assert st[0] == 1 and st[1] == 0, (v, st)
    st = ops.bf16x3_split_status(torch.tensor([[np.inf, -np.inf, np.nan, 1.0]], device=gpu_device))
    assert st[0] == 0 and st[1] == 3
    # the check agrees with a host model of the split on a large mixed sample
    rng = np.random.default_rng(3)
    w = rng.integers(0, 1 << 32, size=200000, dtype=np.uint64).astype(np.uint32).view(np.float32)
    st = ops.bf16x3_split_status(torch.from_numpy(w.copy()).to(gpu_device))

    def rn_bf16(x):
        u = x.view(np.uint32).astype(np.uint64)
        u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFFFFFF
        return ((u >> 16) << 16).astype(np.uint32).view(np.float32)
    fin = np.isfinite(w)
    with np.errstate(all="ignore"):
        x = w[fin]
        hi = rn_bf16(x); r1 = x - hi; mid = rn_bf16(r1); lo = rn_bf16(r1 - mid)
        back = (lo + mid) + hi
    assert st[1] == int((~fin).sum())
    assert st[0] == int((~(back == x)).sum())
    for fn in (ops.pack_linear_bf16x3,):
        with pytest.raises(Exception, match="not exactly hi \\+ mid \\+ lo"):
            fn(torch.tensor([[1.0, float(bad[0])]], device=gpu_device))
    with pytest.raises(Exception, match="non-finite"):
        ops.pack_conv3x3_bf16x3(torch.full((32, 32, 3, 3), float("nan"), device=gpu_device))


@pytest.mark.parametrize("family", ["bf16x3"])
def test_one_hot_spike_gemm_returns_the_weights_bitwise(ops, gpu_device, family):
    """row m of the spike matrix has exactly bit k(m) set: cur[m][:] must be W[:, k(m)] bit for bit - the matrix cores multiply each
    plane by 1.0 and add three exact terms.  The sample contains weights whose low / middle planes are bf16 SUBNORMALS: if
    v_mfma_f32_16x16x32_bf16 flushed subnormal operands those columns would come back with their low bits missing."""
    K, N = 512, 192
    w = _weights_exact_class(K * N, 7).reshape(N, K)
    wt = torch.from_numpy(w.copy()).to(gpu_device)
    sub = ops.bf16x3_split_status(wt)[2]
    assert sub > 100                                                               # the probe is in the sample
    M = K
    z = np.zeros((1, M, K), dtype=np.uint8)
    z[0, np.arange(M), np.arange(K)] = 1
    a = dense_to_planes(z)[0].to(gpu_device)
    cur = ops.spike_gemm_bf16x3(a, K, N, ops.pack_linear_bf16x3(wt)).cpu().numpy()[:, :N]      # [M, N]
    exp = w.T                                                                      # row m = W[:, m]
    nz = exp != 0
    assert np.array_equal(_bits(cur)[nz], _bits(exp)[nz]), "the bf16 matrix cores do not return fp32 weights bit for bit"
    assert np.all(cur[~nz] == 0)
    # two spikes per row: the fp32 sum of two weights, exactly rounded once per addition (values chosen to add exactly)
    w2 = np.zeros((32, 64), dtype=np.float32)
    w2[:, 0] = np.float32(2.0) ** np.arange(-60, -28, dtype=np.float32)
    w2[:, 1] = w2[:, 0] * np.float32(2.0 ** -20)
    z = np.zeros((1, 4, 64), dtype=np.uint8)
    z[0, :, 0] = 1; z[0, :, 1] = 1
    cur = ops.spike_gemm_bf16x3(dense_to_planes(z)[0].to(gpu_device), 64, 32, ops.pack_linear_bf16x3(torch.from_numpy(w2).to(gpu_device)))
    assert np.array_equal(_bits(cur.cpu().numpy()[0, :32]), _bits(w2[:, 0] + w2[:, 1]))


def test_modules_fall_back_loudly_on_unsplittable_weights(gpu_device):
    """a weight the three planes cannot carry routes the head to "f32_strict" (fp32 matrix cores, fp32 VALU heads) with ONE RuntimeWarning
    per weight version; results then equal the explicit precision="f32_strict" run bit for bit and the oracle within tolerance"""
    import snn_automotive_object_detection_amd as S
    torch.manual_seed(0)
    m = S.RPNHeadSNN(32, 3, 6).to(gpu_device)
    feats = [torch.randn(2, 32, 9, 11, device=gpu_device) * 3]
    with torch.no_grad():
        m.shared_conv.weight[3, 5, 1, 1] = 1e-44                                   # an fp32 subnormal: bits below 2^-133
    with pytest.warns(RuntimeWarning, match="f32_strict"):
        a = m(feats)
    with warnings.catch_warnings():
        warnings.simplefilter("error")                                             # second call: cached verdict, no second warning
        a2 = m(feats)
    m2 = S.RPNHeadSNN(32, 3, 6).to(gpu_device)
    m2.load_state_dict(m.state_dict())
    m2.precision = "f32_strict"
    b = m2(feats)
    for x, y, x2 in zip(a[0] + a[1], b[0] + b[1], a2[0] + a2[1]):
        assert torch.equal(x, y) and torch.equal(x, x2)
    o_l, o_b = OR.rpn_head_forward([f.cpu() for f in feats], m.shared_conv.weight.detach().cpu(), m.conv_cls.weight.detach().cpu(),
                                   m.conv_bbox.weight.detach().cpu(), 6)
    bad = sum(int(((g.cpu() - e).abs() > 1e-4).any(1).sum()) for g, e in zip(a[0] + a[1], o_l + o_b))
    assert bad <= 2
    # a healthy weight again (new version): back to bf16x3, silently
    with torch.no_grad():
        m.shared_conv.weight[3, 5, 1, 1] = 0.01
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        m(feats)
    assert m._resolve_precision() == "bf16x3"
    # detector head: a NaN in a LI-head weight (the LI kernels split on the fly) -> strict as well
    d = S.FastRCNNPredictorSNNFull(8 * 49, 64, 5, 6).to(gpu_device)
    with torch.no_grad():
        d.cls_score.weight[1, 2] = float("inf")
    with pytest.warns(RuntimeWarning, match="non-finite"):
        d(torch.randn(10, 8, 7, 7, device=gpu_device))
    assert d._resolve_precision() == "f32_strict"


def test_f32_strict_equals_oracle_like_the_other_families(gpu_device):
    """precision="f32_strict" on a golden-size problem: same tolerance as "f32" (its big contractions ARE the f32 family's; the LI heads
    run the fp32 VALU kernel)"""
    import snn_automotive_object_detection_amd as S
    from oracle import fixtures as FX
    spec = FX.DET_SPECS["det_K9_T12"]
    x, w6, w7, wc, wb = FX.det_inputs(spec)
    d = S.FastRCNNPredictorSNNFull(spec["C"] * 49, spec["Hd"], spec["K"], spec["T"]).to(gpu_device)
    d.load_state_dict({"fc6.weight": w6, "fc7.weight": w7, "cls_score.weight": wc, "bbox_pred.weight": wb})
    d.precision = "f32_strict"
    cls, deltas = d(x.to(gpu_device))
    o_c, o_d = OR.det_head_forward(x, w6, w7, wc, wb, spec["T"])
    rows_bad = int((((cls.cpu() - o_c).abs() > 1e-4).any(1) | ((deltas.cpu() - o_d).abs() > 1e-4).any(1)).sum())
    assert rows_bad <= 1


# ---- (b) non-finite features -------------------------------------------------------------------------------------------------
SPECIALS = [np.inf, -np.inf, np.nan, 3e38, -0.0, 3.4028235e38, 2.5000002, 2.5]


def _documented_trains(T: int) -> np.ndarray:
    """the product's DEFAULT encoders on SPECIALS: the oracle's trains, except +inf = a period-1 neuron"""
    z = OR.encoder_spikes(torch.tensor(SPECIALS, dtype=torch.float32), T).numpy()          # [T, n]
    z[:, 0] = 1.0
    return z


@pytest.mark.parametrize("generic", [False, True])
def test_encoders_on_non_finite_features(ops, gpu_device, monkeypatch, generic):
    T = 8
    oracle = OR.encoder_spikes(torch.tensor(SPECIALS, dtype=torch.float32), T).numpy()
    assert oracle[:, 0].tolist() == [1.0] + [0.0] * (T - 1)                                # the reference: +inf spikes once, then NaN
    expect = oracle if generic else _documented_trains(T)
    if generic:
        monkeypatch.setenv("SNN_ENC_GENERIC", "1")
    p = ops.make_params(ops.LIFParameters(v_th=torch.tensor(0.25)), ops.LIFParameters(alpha=100, v_th=torch.tensor(0.1)))
    D = 64
    x = torch.zeros(3, D)
    x[1, : len(SPECIALS)] = torch.tensor(SPECIALS)
    x[2, D - len(SPECIALS):] = torch.tensor(SPECIALS)
    got = planes_to_dense(ops.encode_rows(x.to(gpu_device), T, p), D)                      # [T, 3, D]
    assert np.array_equal(got[:, 1, : len(SPECIALS)], expect) and np.array_equal(got[:, 2, D - len(SPECIALS):], expect)
    assert got[:, 0].sum() == 0
    f = torch.zeros(1, 32, 4, 5)
    f[0, : len(SPECIALS), 2, 3] = torch.tensor(SPECIALS)
    got = planes_to_dense(ops.encode_nchw(f.to(gpu_device), T, p), 32).reshape(T, 4, 5, 32)
    assert np.array_equal(got[:, 2, 3, : len(SPECIALS)], expect)
    if not generic:                                                                        # period planes: recurrence form and threshold form
        for quant in ("1", "0"):
            monkeypatch.setenv("SNN_STAGE_PERIODS", "1")
            monkeypatch.setenv("SNN_ENC_QUANT", quant)
            e = planes_to_dense(ops.encode_rows(x.to(gpu_device), T, p), D)[:, 1, : len(SPECIALS)]     # e_n planes, n = 1 .. T
            assert (e.sum(axis=0) <= 1).all(), "period planes must be disjoint"
            z = np.zeros_like(e)
            for t in range(T):
                for n in range(1, T + 1):
                    if (t + 1) % n == 0:
                        z[t] = np.maximum(z[t], e[n - 1])
            assert np.array_equal(z, expect), quant
            monkeypatch.delenv("SNN_STAGE_PERIODS")


@pytest.mark.parametrize("generic", [False, True])
def test_heads_on_non_finite_features(gpu_device, monkeypatch, generic):
    """whole heads: with +inf replaced by a huge finite value (documented: +inf = "fires every step") the default path equals the
    oracle; with SNN_ENC_GENERIC=1 it equals the oracle on the ORIGINAL features"""
    import snn_automotive_object_detection_amd as S
    torch.manual_seed(5)
    m = S.RPNHeadSNN(32, 3, 8).to(gpu_device)
    d = S.FastRCNNPredictorSNNFull(8 * 49, 64, 5, 8).to(gpu_device)
    with torch.no_grad():
        m.shared_conv.weight.mul_(4.0)
    f = torch.randn(2, 32, 10, 12) * 2
    x = torch.randn(24, 8, 7, 7) * 2
    sp = torch.tensor(SPECIALS[:5])
    f[0, :5, 4, 6] = sp; f[1, 7:12, 0, 0] = sp
    x[3, 0, 0, :5] = sp; x[20, 7, 6, 2:7] = sp
    if generic:
        monkeypatch.setenv("SNN_ENC_GENERIC", "1")
        f_o, x_o = f, x
    else:
        f_o, x_o = torch.where(torch.isposinf(f), torch.full_like(f, 3e38), f), torch.where(torch.isposinf(x), torch.full_like(x, 3e38), x)
    lg, bb = m([f.to(gpu_device)])
    c, b = d(x.to(gpu_device))
    o_l, o_b = OR.rpn_head_forward([f_o], m.shared_conv.weight.detach().cpu(), m.conv_cls.weight.detach().cpu(), m.conv_bbox.weight.detach().cpu(), 8)
    o_c, o_d = OR.det_head_forward(x_o, d.fc6.weight.detach().cpu(), d.fc7.weight.detach().cpu(), d.cls_score.weight.detach().cpu(),
                                   d.bbox_pred.weight.detach().cpu(), 8)
    for g, e in ((lg[0], o_l[0]), (bb[0], o_b[0]), (c, o_c), (b, o_d)):
        assert torch.isfinite(g).all()
        assert int(((g.cpu() - e).abs() > 1e-4).sum()) <= 2
    if not generic:                                                                        # and the divergence is real: the reference's +inf train is different
        r_l, _ = OR.rpn_head_forward([f], m.shared_conv.weight.detach().cpu(), m.conv_cls.weight.detach().cpu(), m.conv_bbox.weight.detach().cpu(), 8)
        assert float((r_l[0] - o_l[0]).abs().max()) > 1e-4
