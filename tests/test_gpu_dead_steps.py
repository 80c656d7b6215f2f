"""Dead time steps (csrc/snn_kernels.hip: lif_windows).  Norse's LIF cell integrates the input of step t AFTER that step's
membrane update (/root/reference/rpn.py:106, faster_rcnn.py:499,501 -> lif_feed_forward_step), so the 3x3 conv of the last
RPN step, fc6 of the last two and fc7 of the first and last detector step cannot reach any output.  The kernels skip them by
default; SNN_DEAD_STEPS=keep forms the currents of every step.  Both must give the same bits everywhere: outputs, rate
tensors, spike counts and the stage-level spike planes - over all three precisions, both tile shapes and T down to 1.
The oracle-side statement of the same fact (zeroing the dead contractions changes nothing) is tests/test_oracle_kat.py."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _keep_vs_default(monkeypatch, fn):
    # (SNN_DEAD_STEPS=keep runs the all-dense conv launch; the default would put the sparse period planes on the structured-sparse
    # instruction - another fp32 summation order, tests/test_gpu_sparse.py.  The statement here is about the dense kernels' windows.)
    monkeypatch.setenv("SNN_SPARSE", "0")
    monkeypatch.setenv("SNN_DEAD_STEPS", "keep")
    keep = fn()
    monkeypatch.delenv("SNN_DEAD_STEPS")
    trimmed = fn()
    assert len(keep) == len(trimmed)
    for i, (a, b) in enumerate(zip(keep, trimmed)):
        assert torch.equal(a, b), "output %d differs between SNN_DEAD_STEPS=keep and the default" % i
    return keep


@pytest.mark.parametrize("precision", ["bf16x3", "f32", "mxfp6"])
@pytest.mark.parametrize("C,T,shapes", [
    (256, 8, [(2, 48, 96), (2, 24, 48), (2, 12, 24), (2, 6, 12), (2, 3, 6)]),      # the Cityscapes pyramid at 1/4 size
    (128, 12, [(1, 9, 14), (3, 5, 7), (1, 1, 1)]),
    (128, 1, [(2, 7, 9)]),                                                         # T = 1: nothing but step 0
    (128, 2, [(2, 7, 9)]),
    (128, 3, [(1, 13, 5)]),
    (256, 16, [(1, 20, 20)]),
    (256, 24, [(1, 11, 13)]),
])
def test_rpn_head_dead_steps(gpu_device, monkeypatch, precision, C, T, shapes):
    import snn_automotive_object_detection_amd as S
    torch.manual_seed(C + T)
    m = S.RPNHeadSNN(C, 3, T).to(gpu_device)
    m.precision = precision
    with torch.no_grad():
        m.shared_conv.weight.mul_(5.0)                          # let the shared LIF fire
    feats = [torch.randn(n, C, h, w, device=gpu_device) * 1.5 for n, h, w in shapes]

    def run():
        m.spike_rates = False
        lg, bb = m(feats)
        out = [x.clone() for x in lg + bb]
        m.spike_rates = True
        lg, bb, rates = m(feats)
        return out + [x.clone() for x in lg + bb + list(rates)] + [m.last_spike_counts.clone()]
    keep = _keep_vs_default(monkeypatch, run)
    if T >= 4:
        assert int(keep[-1].sum()) > 0


@pytest.mark.parametrize("tile", [("2", "4"), ("2", "2"), ("1", "4"), ("1", "3")])
@pytest.mark.parametrize("T", [5, 8, 12])
def test_rpn_head_dead_steps_tile_shapes(gpu_device, monkeypatch, tile, T):
    import snn_automotive_object_detection_amd as S
    torch.manual_seed(T)
    m = S.RPNHeadSNN(96, 3, T).to(gpu_device)
    with torch.no_grad():
        m.shared_conv.weight.mul_(5.0)
    m.spike_rates = True
    feats = [torch.randn(n, 96, h, w, device=gpu_device) * 1.5 for n, h, w in [(2, 17, 29), (1, 5, 3)]]
    monkeypatch.setenv("SNN_BF16X3_WN", tile[0])
    monkeypatch.setenv("SNN_BF16X3_MT", tile[1])

    def run():
        lg, bb, rates = m(feats)
        return [x.clone() for x in lg + bb + list(rates)] + [m.last_spike_counts.clone()]
    _keep_vs_default(monkeypatch, run)


def test_rpn_head_dead_steps_register_fused_variant(gpu_device, monkeypatch):
    import snn_automotive_object_detection_amd as S
    torch.manual_seed(5)
    m = S.RPNHeadSNN(64, 3, 6).to(gpu_device)
    with torch.no_grad():
        m.shared_conv.weight.mul_(5.0)
    m.spike_rates = True
    feats = [torch.randn(2, 64, 9, 21, device=gpu_device) * 1.5]
    monkeypatch.setenv("SNN_BF16X3_LIF", "reg")

    def run():
        lg, bb, rates = m(feats)
        return [x.clone() for x in lg + bb + list(rates)] + [m.last_spike_counts.clone()]
    reg = _keep_vs_default(monkeypatch, run)
    monkeypatch.delenv("SNN_BF16X3_LIF")
    tile = run()
    for a, b in zip(reg, tile):
        assert torch.equal(a, b)


@pytest.mark.parametrize("precision", ["bf16x3", "f32", "mxfp6"])
@pytest.mark.parametrize("R,D_ch,Hd,K,T", [(300, 128, 128, 9, 12), (37, 128, 128, 5, 8), (5, 128, 128, 3, 1), (5, 128, 128, 3, 2),
                                          (64, 128, 128, 3, 3), (513, 128, 256, 11, 4), (200, 128, 128, 9, 24), (45, 128, 128, 2, 16)])
def test_det_head_dead_steps(gpu_device, monkeypatch, precision, R, D_ch, Hd, K, T):
    import snn_automotive_object_detection_amd as S
    torch.manual_seed(R + T)
    m = S.FastRCNNPredictorSNNFull(D_ch * 49, Hd, K, T).to(gpu_device)
    m.precision = precision
    with torch.no_grad():
        m.fc7.weight.mul_(3.0)                                  # let lif7 fire
    x = torch.randn(R, D_ch, 7, 7, device=gpu_device) * 1.5

    def run():
        m.spike_rates = False
        c, b = m(x)
        out = [c.clone(), b.clone()]
        m.spike_rates = True
        rates = m(x)
        return out + [r.clone() for r in rates] + [t.clone() for t in m.last_spike_counts]
    keep = _keep_vs_default(monkeypatch, run)
    if T >= 8:
        assert int(keep[-1].sum()) > 0 and int(keep[-2].sum()) > 0          # lif6 and lif7 both fire


@pytest.mark.parametrize("tile", [("2", "4"), ("2", "3"), ("1", "4"), ("1", "2")])
def test_det_head_dead_steps_tile_shapes(gpu_device, monkeypatch, tile):
    import snn_automotive_object_detection_amd as S
    torch.manual_seed(11)
    m = S.FastRCNNPredictorSNNFull(32 * 49, 160, 9, 12).to(gpu_device)
    with torch.no_grad():
        m.fc7.weight.mul_(3.0)
    x = torch.randn(333, 32, 7, 7, device=gpu_device) * 1.5
    monkeypatch.setenv("SNN_BF16X3_WN", tile[0])
    monkeypatch.setenv("SNN_BF16X3_MT", tile[1])

    def run():
        m.spike_rates = False
        c, b = m(x)
        m.spike_rates = True
        rates = m(x)
        return [c.clone(), b.clone()] + [r.clone() for r in rates] + [t.clone() for t in m.last_spike_counts]
    _keep_vs_default(monkeypatch, run)


def test_det_head_nonzero_rest_potential_keeps_step_zero(gpu_device, monkeypatch):
    """v_leak > v_th: lif6 fires at step 0, so fc7's current of step 0 is NOT zero and its window must start at step 0.  The
    modules refuse a non-default rest potential (ops.make_params: the LI heads assume the default), so this goes through the
    C ABI wrappers with hand-made parameters; the comparison is keep-all-steps against the default, bit for bit."""
    import snn_automotive_object_detection_amd as S
    from snn_automotive_object_detection_amd import ops
    torch.manual_seed(3)
    T = 6
    m = S.FastRCNNPredictorSNNFull(32 * 49, 64, 5, T).to(gpu_device)
    x = torch.randn(40, 32, 7, 7, device=gpu_device)
    w6, w7, wh = m._packed(inner=0)
    for v_leak, fires_at_0 in ((0.2, True), (0.05, False)):
        p = m._params()
        p.v_leak = v_leak                                        # 0.2 > v_th_lif = 0.1: every lif6 neuron fires at step 0
        planes = ops.encode_rows(x.flatten(1), T, m._params())
        spk = _keep_vs_default(monkeypatch, lambda: [ops.spike_gemm_lif_bf16x3(planes, 32 * 49, 64, p, ops.pack_linear_bf16x3(m.fc6.weight)).clone()])[0]
        assert bool(spk[0].ne(0).any()) == fires_at_0

        def run():
            out = []
            for rates in (False, True):
                c, b, extras = ops.det_head_forward(x, 64, 5, 20, T, p, w6, w7, wh, spike_rates=rates)
                out += [c.clone(), b.clone()] + ([e.clone() for e in extras] if rates else [])
            return out
        _keep_vs_default(monkeypatch, run)


def test_stage_level_planes_are_complete(gpu_device, monkeypatch):
    """the public stage entry points return EVERY spike plane (their window is steps 0 .. T-2): conv+LIF and linear+LIF"""
    from snn_automotive_object_detection_amd import ops
    import snn_automotive_object_detection_amd as S
    torch.manual_seed(9)
    T, C = 7, 64
    m = S.RPNHeadSNN(C, 3, T).to(gpu_device)
    with torch.no_grad():
        m.shared_conv.weight.mul_(5.0)
    p = m._params()
    shapes = [(2, 10, 13)]
    enc = ops.encode_nchw(torch.randn(2, C, 10, 13, device=gpu_device) * 1.5, T, p)
    w = ops.pack_conv3x3_bf16x3(m.shared_conv.weight)
    spk = _keep_vs_default(monkeypatch, lambda: [ops.conv3x3_lif_bf16x3(enc, shapes, C, C, p, w).clone()])[0]
    assert int(spk[T - 1].ne(0).sum()) > 0
    a = torch.randint(-2 ** 31, 2 ** 31 - 1, (T, 50, 4), dtype=torch.int64, device=gpu_device).to(torch.int32)
    w6 = ops.pack_linear_bf16x3(torch.randn(96, 128, device=gpu_device) * 0.05)
    _keep_vs_default(monkeypatch, lambda: [ops.spike_gemm_lif_bf16x3(a, 128, 96, p, w6).clone()])
