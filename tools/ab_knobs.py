"""A/B of run-time knobs of libsnnhip.so on ONE box, interleaved rounds, on the bench's own backbone-fed inputs (operand
statistics set the clock the chip holds under the matrix-core kernels, so random tensors must not be used to rank builds):
  python tools/ab_knobs.py "SNN_DEAD_STEPS=keep" "" "SNN_BF16X3_WN=1" ...     ("" = defaults)
Prints conv+LIF launch, RPN head, detector head (ms, best and median of the rounds) per setting."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                                     # noqa: E402
import bench                                                     # noqa: E402
import snn_automotive_object_detection_amd as S                  # noqa: E402
from snn_automotive_object_detection_amd import _lib, ops        # noqa: E402


def main():
    settings = sys.argv[1:] or ["SNN_DEAD_STEPS=keep", ""]
    wl_name = os.environ.get("AB_WORKLOAD", "cityscapes")
    rounds, iters = int(os.environ.get("AB_ROUNDS", "5")), int(os.environ.get("AB_ITERS", "20"))
    dev = torch.device("cuda:0")
    wl = dict(bench.WORKLOADS[wl_name])
    if os.environ.get("AB_T_RPN"):
        wl["T_rpn"] = int(os.environ["AB_T_RPN"])
    if os.environ.get("AB_T_DET"):
        wl["T_det"] = int(os.environ["AB_T_DET"])
    torch.manual_seed(4321)
    model = S.create_model(wl["dataset"], wl["K"], True, True, 0, False, False, 8, 12).to(dev).eval()
    leg = bench.Leg(wl, "bf16x3", dev, 1000, "backbone", model)
    del model
    res = {s: {"conv": [], "rpn": [], "det": []} for s in settings}

    def apply(setting):
        for kv in filter(None, setting.split(",")):
            k, v = kv.split("=")
            os.environ[k] = v
        _lib.reload_knobs()

    def clear(setting):
        for kv in filter(None, setting.split(",")):
            os.environ.pop(kv.split("=")[0], None)
        _lib.reload_knobs()

    p = leg.rpn_head._params()
    w_sh = leg.rpn_head._packed_shared()
    leg.rpn_head(leg.feats)
    w_hd = leg.rpn_head._cache_heads.val
    T = wl["T_rpn"]
    for r in range(rounds + 1):
        for s in settings:
            apply(s)
            try:
                ops.rpn_head_forward(leg.feats, bench.C, bench.A, T, p, w_sh, w_hd, stage_mask=7)
                conv = leg.time_ms(lambda: ops.rpn_head_forward(leg.feats, bench.C, bench.A, T, p, w_sh, w_hd, stage_mask=2), iters)
                rpn = leg.time_ms(lambda: leg.rpn_head(leg.feats), iters)
                det = leg.time_ms(lambda: leg.det_head(leg.rois), iters)
            finally:
                clear(s)
            if r:                                                # round 0 = warm-up
                res[s]["conv"].append(conv); res[s]["rpn"].append(rpn); res[s]["det"].append(det)
    for s in settings:
        print("%-40s conv+LIF %.4f / %.4f ms   rpn head %.4f / %.4f ms   det head %.4f / %.4f ms   (best / median of %d rounds x %d)" % (
            s or "(defaults)", min(res[s]["conv"]), statistics.median(res[s]["conv"]), min(res[s]["rpn"]), statistics.median(res[s]["rpn"]),
            min(res[s]["det"]), statistics.median(res[s]["det"]), rounds, iters))


if __name__ == "__main__":
    main()
