"""Per-work-group timeline of k_gemm_lif_sparse (a -DSNN_EXP_TIMELINE build: bash tools/ab_build.sh TL:"-DSNN_EXP_TIMELINE", then on the
GPU box SNN_HIP_LIB=tools/_ab/lib_TL.so python tools/sparse_timeline.py [fc6]): s_memrealtime stamps at entry / K-loop start / K-loop end /
end of the first epilogue pass / exit.  Run on the bench's own pyramid (default: the RPN conv) or RoI features (`fc6`: the detector's fc6,
round 5 - VERDICT r4 item 3)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                                     # noqa: E402
import bench                                                     # noqa: E402
import snn_automotive_object_detection_amd as S                  # noqa: E402
from snn_automotive_object_detection_amd import _lib, ops        # noqa: E402

dev = torch.device("cuda:0")
wl = dict(bench.WORKLOADS["cityscapes"])
torch.manual_seed(4321)
model = S.create_model(wl["dataset"], wl["K"], True, True, 0, False, False, 8, 12).to(dev).eval()
leg = bench.Leg(wl, "bf16x3", dev, 1000, "backbone", model)
del model
lib = _lib.load()


def report(raw, slots, what):
    raw = raw[raw[:, 7] != 0]
    t = raw[:, :5].double() / 100.0                              # us
    t0 = float(t[:, 0].min())
    print("%s: work-groups stamped: %d; launch span %.1f us" % (what, raw.shape[0], float(t[:, 4].max()) - t0))
    print("mean us per work-group: before K loop %.2f | K loop %.2f | epilogue pass 0 %.2f | epilogue pass 1 %.2f | whole %.2f" % (
        float((t[:, 1] - t[:, 0]).mean()), float((t[:, 2] - t[:, 1]).mean()), float((t[:, 3] - t[:, 2]).mean()), float((t[:, 4] - t[:, 3]).mean()),
        float((t[:, 4] - t[:, 0]).mean())))
    img = raw[:, 7].double() / 100.0
    if float((t[:, 4] - t[:, 3]).mean()) < 0.2:                  # FAT shapes with the LIF in registers: one phase (stamps 7 and 3 sit at its end)
        print("epilogue = LIF in registers + spike stores (no tile image, no barrier): %.2f us" % float((t[:, 3] - t[:, 2]).mean()))
    else:
        print("epilogue pass 0: accumulators -> tile image (two barriers) %.2f us | LIF recurrence + spike stores %.2f us" % (
            float((img - t[:, 2]).mean()), float((t[:, 3] - img).mean())))
    cu = (raw[:, 6] << 16) | (raw[:, 5] & 0x0000ff00) | ((raw[:, 5] >> 13) & 0x7)    # xcc | cu_id / sh / se bits
    ids, counts = torch.unique(cu, return_counts=True)
    print("distinct (XCC, CU) slots seen: %d; work-groups per slot: min %d max %d" % (ids.numel(), int(counts.min()), int(counts.max())))
    busy = float((t[:, 4] - t[:, 0]).sum()) / slots
    print("sum of work-group lifetimes / %d slots = %.1f us (= the span if every slot were always occupied)" % (slots, busy))
    # start / end skew: how long does the launch run with fewer than all slots busy
    starts, ends = t[:, 0] - t0, t[:, 4] - t0
    print("work-group starts: first round (<= %d) all started by %.1f us; last start %.1f us; first exit %.1f us; last exit %.1f us" % (
        slots, float(starts.sort().values[min(slots, starts.numel()) - 1]), float(starts.max()), float(ends.min()), float(ends.max())))
    kl = (t[:, 2] - t[:, 1])
    print("K loop per work-group: min %.2f  median %.2f  max %.2f us" % (float(kl.min()), float(kl.median()), float(kl.max())))
    return t


if len(sys.argv) > 1 and sys.argv[1] == "fc6":
    T, R, D, HD = wl["T_det"], int(leg.rois.shape[0]), bench.C * 49, bench.HD
    leg.det_head(leg.rois)
    torch.cuda.synchronize()
    assert lib.snn_debug_last_fc6_path() == 1
    import ctypes as Ct
    o12 = (Ct.c_int32 * 12)()
    assert lib.snn_debug_tile_shape(0, R, D, HD, T, 0, 6, o12) == 0 and o12[8] == 1
    n_wg, steps = int(o12[5]), D // 64
    al = lambda x: (x + 255) // 256 * 256
    Dw = D // 32
    o_cur = al(T * R * Dw * 4)                                   # det_ws_layout: the encoder planes, then the currents region = the sparse side buffers
    side = al(max(T - 3, 1) * (Dw // 2) * 4 * R * 4) + al(R * 4) + 256
    ws = ops._WS.get(dev, 1)
    ws[o_cur + side: o_cur + side + n_wg * 64].zero_()           # (the allocator hands out used memory: stale bytes would read as stamps)
    for _ in range(3):
        leg.det_head(leg.rois)
    torch.cuda.synchronize()
    raw = ws[o_cur + side: o_cur + side + n_wg * 64].view(torch.int64).view(-1, 8).cpu()
    t = report(raw, 512, "k_gemm_lif_sparse<false, %d> (fc6, T = %d: %d period planes, %d RoIs per tile, %d work-groups, %d K steps of 64)" % (
        o12[7], T, o12[4], o12[3], n_wg, steps))
    print("K loop: %.3f us per 64-k step (mean)" % (float((t[:, 2] - t[:, 1]).mean()) / steps))
    sys.exit(0)

T, C, A = wl["T_rpn"], bench.C, bench.A
P = sum(f.shape[0] * f.shape[2] * f.shape[3] for f in leg.feats)
lv = (_lib.snn_rpn_level * len(leg.feats))(*[_lib.snn_rpn_level(f.data_ptr(), f.shape[0], f.shape[2], f.shape[3], 0) for f in leg.feats])
p = leg.rpn_head._params()
ws_bytes = lib.snn_rpn_head_workspace_bytes(lv, len(leg.feats), C, A, T, p.precision)
n_wg = 4 * ((P + 63) // 64) + 64
ws = ops._WS.get(dev, ws_bytes + n_wg * 64 + (64 << 20))
off = ws_bytes                                                  # the stamps land behind everything the head needs of its workspace
w_sh, w_hd = leg.rpn_head._packed_shared(), None
leg.rpn_head(leg.feats)
w_hd = leg.rpn_head._cache_heads.val
ws[off: off + n_wg * 64].zero_()
for _ in range(5):
    ops.rpn_head_forward(leg.feats, C, A, T, p, w_sh, w_hd, stage_mask=2)
torch.cuda.synchronize()
assert lib.snn_debug_last_conv_path() == 1
raw = ws[off: off + n_wg * 64].view(torch.int64).view(-1, 8).cpu()
report(raw, 512, "k_gemm_lif_sparse<true, 1> (RPN conv, T = %d)" % T)
