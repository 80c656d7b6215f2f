"""Interval timeline of the ping-pong conv (k_conv_lif_pp, csrc/snn_sparse_pp.h) from a -DSNN_EXP_TIMELINE build:
  bash tools/ab_build.sh TLPP:"-DSNN_PINGPONG -DSNN_EXP_TIMELINE -DSNN_CONV_PP_DEFAULT=1"   then on the GPU box   SNN_HIP_LIB=tools/_ab/lib_TLPP.so python tools/pp_timeline.py
Work-group 0 stamps s_memtime (shader-clock cycles) five times per step and half on its first tile pair: Y start | Y end | barrier passed (X start) |
matrix instructions issued | copies of the next step landed (then the barrier behind X).  Printed per half: mean cycles of each segment."""
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
T, C, A = wl["T_rpn"], bench.C, bench.A
lv = (_lib.snn_rpn_level * len(leg.feats))(*[_lib.snn_rpn_level(f.data_ptr(), f.shape[0], f.shape[2], f.shape[3], 0) for f in leg.feats])
p = leg.rpn_head._params()
ws_bytes = lib.snn_rpn_head_workspace_bytes(lv, len(leg.feats), C, A, T, p.precision)
ws = ops._WS.get(dev, ws_bytes + (64 << 20))
off = ws_bytes                                                  # the stamps land behind everything the head needs of its workspace
w_sh = leg.rpn_head._packed_shared()
leg.rpn_head(leg.feats)
w_hd = leg.rpn_head._cache_heads.val
ws[off: off + 2 * 64 * 64].zero_()
for _ in range(5):
    ops.rpn_head_forward(leg.feats, C, A, T, p, w_sh, w_hd, stage_mask=2)
torch.cuda.synchronize()
raw = ws[off: off + 2 * 64 * 64].view(torch.int64).view(2, 64, 8).cpu().double()
n = 36
for h in range(2):
    t = raw[h, :n, :5]
    if float(t.abs().sum()) == 0:
        print("half %d: no stamps (is this a -DSNN_EXP_TIMELINE build with SNN_CONV_PP on?)" % h)
        continue
    y = t[:, 1] - t[:, 0]; b1 = t[:, 2] - t[:, 1]; x = t[:, 3] - t[:, 2]; w = t[:, 4] - t[:, 3]
    b2 = t[1:, 0] - t[:-1, 4]
    step = t[1:, 0] - t[:-1, 0]
    sl = slice(2, n - 2)
    print("half %d (cycles, mean over steps 2..%d): Y %.0f | barrier behind Y %.0f | X (matrix instructions issued) %.0f | wait for the next copies %.0f | barrier behind X %.0f | step %.0f" % (
        h, n - 3, float(y[sl].mean()), float(b1[sl].mean()), float(x[sl].mean()), float(w[sl].mean()), float(b2[sl].mean()), float(step[sl].mean())))
    print("   per step X:", " ".join("%d" % v for v in x.tolist()))
    print("   per step Y:", " ".join("%d" % v for v in y.tolist()))
    print("   per step wait:", " ".join("%d" % v for v in w.tolist()))
print("tile (steps 0..35): half 0 %.0f cycles, half 1 %.0f cycles; half 1 starts %.0f cycles behind half 0" % (
    float(raw[0, n - 1, 4] - raw[0, 0, 0]), float(raw[1, n - 1, 4] - raw[1, 0, 0]), float(raw[1, 0, 0] - raw[0, 0, 0])))
