"""GPU parity of the fused MultiScaleRoIAlign + encoder kernel: pooled values BIT-IDENTICAL to the
CPU restatement of torchvision's roi_align (oracle/roi_align_oracle.py: same fp32 operations in the same order), encoder
planes bit-exact on those pooled values, the committed fixture tests/golden/roialign_c8.npz, and the fused detector head
against the two-step path."""
import numpy as np
import pytest
import torch

from oracle import fixtures as FX
from oracle import roi_align_oracle as RA
from oracle import snn_oracle as OR
from tests._util import planes_to_dense

pytestmark = pytest.mark.gpu


def _setup(dev, R=300, C=32, seed=0):
    from snn_automotive_object_detection_amd.stock.roi_align import MultiScaleRoIAlign
    g = torch.Generator().manual_seed(seed)
    sizes = [(96, 160), (48, 80), (24, 40), (12, 20)]
    feats = {str(i): torch.randn((2, C, h, w), generator=g).to(dev) for i, (h, w) in enumerate(sizes)}
    feats["pool"] = torch.randn((2, C, 6, 10), generator=g).to(dev)
    boxes = []
    for n in range(2):
        xy = torch.rand((R // 2, 2), generator=g) * torch.tensor([600.0, 360.0])
        wh = torch.exp(torch.rand((R // 2, 2), generator=g) * 5.8 + 1.0)          # 3 .. 900 px: all four levels
        b = torch.cat([xy, xy + wh], dim=1)
        b[0] = torch.tensor([-20.0, -10.0, 30.0, 25.0])                          # partly outside
        b[1] = torch.tensor([630.0, 370.0, 700.0, 420.0])                        # beyond the far edge
        b[2] = torch.tensor([100.0, 100.0, 100.2, 100.1])                        # degenerate: clamps to 1 px
        boxes.append(b.to(dev))
    pool = MultiScaleRoIAlign(["0", "1", "2", "3"], 7, 2)
    return pool, feats, boxes, [(384, 640), (384, 640)]


def test_roi_align_encode_matches_stock_op_and_oracle_encoder(gpu_device):
    from snn_automotive_object_detection_amd import ops
    pool, feats, boxes, shapes = _setup(gpu_device)
    T = 12
    p = ops.make_params(ops.LIFParameters(v_th=torch.tensor(0.25)), ops.LIFParameters(alpha=100, v_th=torch.tensor(0.1)))
    flist, scales, rois, lvl = pool.assign(feats, boxes, shapes)
    assert set(lvl.tolist()) == {0, 1, 2, 3}
    planes, pooled = ops.roi_align_encode(flist, scales, rois[:, 1:5], rois[:, 0], lvl, T, p, want_pooled=True)
    # reference = the restatement of torchvision's CPU kernel (IEEE fp32, torchvision's operation order): same bits
    ref = RA.multiscale_roi_align({k: v.cpu() for k, v in feats.items()}, [b.cpu() for b in boxes], shapes).flatten(1)
    assert pooled.shape == ref.shape
    assert np.array_equal(pooled.cpu().numpy(), ref.numpy()), float((pooled.cpu() - ref).abs().max())
    # the stock-torch stand-in on the CPU gives the same bits too; on the GPU torch's own division is not correctly rounded,
    # which moves sample coordinates by an ulp and pooled values by up to ~4e-5 (measured, tools/dbg_roi.py)
    assert torch.equal(pool({k: v.cpu() for k, v in feats.items()}, [b.cpu() for b in boxes], shapes).flatten(1), ref)
    ref_gpu = pool(feats, boxes, shapes).flatten(1)
    assert float((pooled - ref_gpu).abs().max()) <= 2e-4 * max(1.0, float(ref.abs().max()))
    # the encoder half is bit-exact on the kernel's own pooled values
    z = OR.encoder_spikes(pooled.cpu(), T)
    assert np.array_equal(planes_to_dense(planes, pooled.shape[1]), z.numpy())


@pytest.mark.parametrize("T,periods", [(12, "1"), (12, "0"), (24, "1"), (32, "1"), (5, "0")])
def test_word_major_kernels_table_driven_and_per_element(gpu_device, monkeypatch, T, periods):
    """the kernels the fused detector head runs (word-major planes [T][Dw][R]): the table-driven one (sample geometry once per wave
    and RoI, 8-byte tap pairs; default) and the per-element one (SNN_ROI_TAB=0) - pooled values bit-identical to the restatement of
    torchvision's kernel, planes identical to the row-major stage kernel's, for spike planes and period planes, at every
    several work-group shapes (SNN_ROI_E element groups x SNN_ROI_RW RoIs per wave)"""
    from snn_automotive_object_detection_amd import ops
    pool, feats, boxes, shapes = _setup(gpu_device, R=333, C=40, seed=5)            # D = 1960: 61.25 plane words (ragged last word)
    p = ops.make_params(ops.LIFParameters(v_th=torch.tensor(0.25)), ops.LIFParameters(alpha=100, v_th=torch.tensor(0.1)))
    flist, scales, rois, lvl = pool.assign(feats, boxes, shapes)
    monkeypatch.setenv("SNN_STAGE_PERIODS", periods)
    ref_planes, ref_pooled = ops.roi_align_encode(flist, scales, rois[:, 1:5], rois[:, 0], lvl, T, p, want_pooled=True)
    want = RA.multiscale_roi_align({k: v.cpu() for k, v in feats.items()}, [b.cpu() for b in boxes], shapes).flatten(1)
    assert np.array_equal(ref_pooled.cpu().numpy(), want.numpy())
    R, Dw = ref_planes.shape[1], ref_planes.shape[2]
    monkeypatch.setenv("SNN_STAGE_PLANES", "wm")
    for tab, e, rw in (("1", "0", "0"), ("0", "0", "0"), ("1", "7", "8"), ("1", "1", "1"), ("1", "3", "5")):
        monkeypatch.setenv("SNN_ROI_TAB", tab)
        monkeypatch.setenv("SNN_ROI_E", e)
        monkeypatch.setenv("SNN_ROI_RW", rw)
        planes, pooled = ops.roi_align_encode(flist, scales, rois[:, 1:5], rois[:, 0], lvl, T, p, want_pooled=True)
        assert np.array_equal(pooled.cpu().numpy(), want.numpy()), (tab, e, rw)
        assert torch.equal(planes.view(T, Dw, R).transpose(1, 2), ref_planes), (tab, e, rw)


def test_table_driven_kernel_on_borders_and_tiny_maps(gpu_device, monkeypatch):
    """clamped columns / rows (the 8-byte tap pair is read one to the left), samples outside the map, 2-pixel-wide levels"""
    from snn_automotive_object_detection_amd import ops
    from snn_automotive_object_detection_amd.stock.roi_align import MultiScaleRoIAlign
    g = torch.Generator().manual_seed(9)
    sizes = [(16, 16), (8, 8), (4, 4), (2, 2)]
    feats = {str(i): torch.randn((1, 8, h, w), generator=g).to(gpu_device) for i, (h, w) in enumerate(sizes)}
    b = torch.tensor([[0.0, 0.0, 64.0, 64.0], [60.0, 60.0, 64.0, 64.0], [63.5, 0.0, 64.0, 64.0], [-30.0, -30.0, 10.0, 10.0],
                      [0.0, 62.0, 64.0, 66.0], [10.0, 10.0, 500.0, 500.0], [63.9, 63.9, 64.0, 64.0], [0.0, 0.0, 3.0, 3.0]])
    boxes = [torch.cat([b, torch.rand((40, 4), generator=g) * 32 + torch.tensor([0.0, 0.0, 32.0, 32.0])]).to(gpu_device)]
    shapes = [(64, 64)]
    pool = MultiScaleRoIAlign(["0", "1", "2", "3"], 7, 2)
    flist, scales, rois, lvl = pool.assign(feats, boxes, shapes)
    p = ops.make_params(ops.LIFParameters(v_th=torch.tensor(0.25)), ops.LIFParameters(alpha=100, v_th=torch.tensor(0.1)))
    want = RA.multiscale_roi_align({k: v.cpu() for k, v in feats.items()}, [x.cpu() for x in boxes], shapes).flatten(1)
    monkeypatch.setenv("SNN_STAGE_PLANES", "wm")
    for tab in ("1", "0"):
        monkeypatch.setenv("SNN_ROI_TAB", tab)
        _, pooled = ops.roi_align_encode(flist, scales, rois[:, 1:5], rois[:, 0], lvl, 6, p, want_pooled=True)
        assert np.array_equal(pooled.cpu().numpy(), want.numpy()), tab


@pytest.mark.parametrize("precision", ["bf16x3", "f32"])
def test_fused_head_equals_two_step_path(gpu_device, precision):
    import snn_automotive_object_detection_amd as S
    pool, feats, boxes, shapes = _setup(gpu_device, R=200, C=16, seed=3)
    torch.manual_seed(1)
    head = S.FastRCNNPredictorSNNFull(16 * 49, 128, 9, 12).to(gpu_device)
    head.precision = precision
    box_features = pool({k: v.cpu() for k, v in feats.items()}, [b.cpu() for b in boxes], shapes).to(gpu_device)
    c_ref, b_ref = head(box_features)
    flist, scales, rois, lvl = pool.assign(feats, boxes, shapes)
    c_fused, b_fused = head.forward_roialign(flist, scales, rois, lvl)
    # pooled values agree to the last bit almost everywhere; a differing ulp can flip an encoder spike at a tie
    rows_off = ((c_fused - c_ref).abs().amax(1) > 1e-4) | ((b_fused - b_ref).abs().amax(1) > 1e-4)
    assert int(rows_off.sum()) <= 2


@pytest.mark.parametrize("wm", ["wm", "rm"])
def test_fused_head_on_the_committed_fixture(gpu_device, monkeypatch, wm):
    """tests/golden/roialign_c8.npz: pooled features of the restatement (bit-identical) and the detector-head oracle's outputs
    on them, against snn_roi_align_encode / snn_det_head_forward_roialign (both plane layouts)"""
    import snn_automotive_object_detection_amd as S
    from snn_automotive_object_detection_amd import ops
    from snn_automotive_object_detection_amd.stock.roi_align import MultiScaleRoIAlign
    monkeypatch.setenv("SNN_PLANES", wm)
    spec = FX.ROI_SPECS["roialign_c8"]
    exp = FX.load_expected("roialign_c8")
    feats, boxes, shapes = FX.roi_inputs(spec)
    feats = {k: v.to(gpu_device) for k, v in feats.items()}
    boxes = [b.to(gpu_device) for b in boxes]
    pool = MultiScaleRoIAlign(["0", "1", "2", "3"], 7, 2)
    flist, scales, rois, lvl = pool.assign(feats, boxes, shapes)
    head = S.FastRCNNPredictorSNNFull(spec["C"] * 49, spec["Hd"], spec["K"], spec["T"]).to(gpu_device)
    w6, w7, wc, wb = FX.roi_head_weights(spec)
    head.load_state_dict({"fc6.weight": w6, "fc7.weight": w7, "cls_score.weight": wc, "bbox_pred.weight": wb})
    _, pooled = ops.roi_align_encode(flist, scales, rois[:, 1:5], rois[:, 0], lvl, spec["T"], head._params(), want_pooled=True)
    assert np.array_equal(pooled.cpu().numpy().reshape(exp["pooled"].shape), exp["pooled"])
    cls, bbox = head.forward_roialign(flist, scales, rois, lvl)
    rows_off = (np.abs(cls.cpu().numpy() - exp["cls"]).max(1) > 1e-4) | (np.abs(bbox.cpu().numpy() - exp["bbox"]).max(1) > 1e-4)
    assert int(rows_off.sum()) == 0, int(rows_off.sum())


def test_roi_heads_uses_fusion_and_matches_unfused_on_identical_pooled_values(gpu_device):
    """RoIHeadsSNN.forward with and without the fusion.  The un-fused path's pooling is moved to the CPU here (bit-identical to
    the kernel's; torch's GPU division would move pooled values by ~4e-5 and flip encoder spikes in a quarter of the RoIs):
    then both paths see the same RoI features and must agree."""
    import snn_automotive_object_detection_amd as S
    torch.manual_seed(0)
    m = S.create_model("cityscapes", 9, True, True, 0, False, False, num_steps_rpn=4, num_steps_detector=6)
    m.transform.min_size, m.transform.max_size = 256, 512
    m = m.to(gpu_device).eval()
    img = [torch.rand((3, 256, 512), device=gpu_device)]
    assert m.roi_heads.fuse_roi_align
    with torch.no_grad():
        il, _ = m.transform(img)
        fm = m.backbone(il.tensors)
        props, _ = m.rpn(il, fm)
        a, _ = m.roi_heads(fm, props, il.image_sizes)
        m.roi_heads.fuse_roi_align = False
        pool = m.roi_heads.box_roi_pool
        stock_forward = pool.forward
        pool.forward = lambda x, boxes, shapes: stock_forward({k: v.cpu() for k, v in x.items()}, [b.cpu() for b in boxes], shapes).to(gpu_device)
        b, _ = m.roi_heads(fm, props, il.image_sizes)
    assert a[0]["all_scores"].shape == b[0]["all_scores"].shape
    assert torch.equal(a[0]["all_scores"], b[0]["all_scores"]) and torch.equal(a[0]["all_boxes"], b[0]["all_boxes"])
    assert torch.equal(a[0]["boxes"], b[0]["boxes"])


@pytest.mark.parametrize("R,C,Hd,K,T", [(300, 64, 128, 5, 12), (1000, 256, 256, 9, 12), (130, 128, 64, 3, 14), (38, 64, 64, 3, 6), (6, 64, 64, 3, 8), (60, 192, 64, 3, 8)])
def test_roialign_encoder_fold_writes_the_same_planes(gpu_device, monkeypatch, R, C, Hd, K, T):
    """round 6 (row f1 on the DEFAULT product path): k_roi_align_encode_perm - RoIAlign + encoder + fc6's reduction order + compression in ONE
    launch - against k_roi_align_encode_tab -> k_permute_planes -> k_compress_planes (SNN_ENC_FOLD=0): the dense planes e_1, e_2 and the
    compressed planes e_3 .. in the workspace bit for bit, the head's outputs and spike counts with them"""
    import snn_automotive_object_detection_amd as S
    from snn_automotive_object_detection_amd import _lib, ops
    pool, feats, boxes, shapes = _setup(gpu_device, R=R, C=C, seed=R)
    torch.manual_seed(R + T)
    head = S.FastRCNNPredictorSNNFull(C * 49, Hd, K, T).to(gpu_device)
    flist, scales, rois, lvl = pool.assign(feats, boxes, shapes)
    Rn = int(rois.shape[0])
    al = lambda v: (v + 255) // 256 * 256
    Dw, Tc = C * 49 // 32, T - 2
    o_cur = al(T * Rn * Dw * 4)                                 # det_ws_layout: the encoder planes (fc6 reads them at the front), then the side buffers
    dense_bytes, cmp_bytes = 2 * Dw * Rn * 4, (Tc - 2) * (Dw // 2) * 4 * Rn * 4

    def run():
        ws = ops._WS.get(gpu_device, 1)
        if ws.numel() >= o_cur + cmp_bytes:
            ws[: o_cur + cmp_bytes].fill_(0x5a)
        c, b = head.forward_roialign(flist, scales, rois, lvl)
        assert _lib.load().snn_debug_last_fc6_path() == 1
        ws = ops._WS.get(gpu_device, 1)
        return (c.clone(), b.clone()), ws[:dense_bytes].clone(), ws[o_cur: o_cur + cmp_bytes].clone()
    run()                                                       # (sizes the workspace)
    monkeypatch.setenv("SNN_ENC_FOLD", "0")
    a, dense_a, cmp_a = run()
    monkeypatch.delenv("SNN_ENC_FOLD")
    for _ in range(2):
        b, dense_b, cmp_b = run()
        assert torch.equal(dense_a, dense_b), int((dense_a != dense_b).sum())
        assert torch.equal(cmp_a, cmp_b), int((cmp_a != cmp_b).sum())
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert int((dense_a != 0).sum()) > 0
    head.spike_rates = True                                     # (window T - 1: one more compressed plane)
    r_fold = head.forward_roialign(flist, scales, rois, lvl)
    c_fold = [c.clone() for c in head.last_spike_counts]
    monkeypatch.setenv("SNN_ENC_FOLD", "0")
    head.forward_roialign(flist, scales, rois, lvl)
    assert all(torch.equal(p, q) for p, q in zip(c_fold, head.last_spike_counts))
