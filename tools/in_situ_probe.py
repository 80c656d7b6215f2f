"""Why is the conv+LIF launch 7-12 % slower inside create_model's forward than stand-alone?
Run with a -DSNN_EXP_CLOCK build (bash tools/ab_build.sh CLK:"-DSNN_EXP_CLOCK", then on the GPU box
  SNN_HIP_LIB=tools/_ab/lib_CLK.so python tools/in_situ_probe.py):
every work-group of the T-in-tile launches stamps s_memtime (shader clock cycles) and s_memrealtime (100 MHz) around its K loop
and writes the differences behind the spike planes.  For the SAME model, weights and images the conv+LIF launch is run
  (a) in situ: model(images), behind the stock backbone,
  (b) heads only: RPN head + detector head back to back on the backbone's features (what bench.py times),
  (c) conv+LIF alone, back to back (bench.py's roofline.launch_ms),
  (d) in situ with the GPU left idle for a few ms in front of every forward,
and reported as: in-kernel clock (GHz), K-loop time per work-group (us), K-loop cycles per work-group, launch time by HIP events
where the launch can be bracketed.  Also the densities of the encoder's period planes of the pyramid."""
import ctypes as C
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                                     # noqa: E402
import snn_automotive_object_detection_amd as S                  # noqa: E402
from snn_automotive_object_detection_amd import _lib, ops        # noqa: E402

dev = torch.device("cuda:0")
T_RPN, T_DET, CH, A = 8, 12, 256, 3
torch.manual_seed(0)
model = S.create_model("cityscapes", 9, True, True, 0, False, False, T_RPN, T_DET).to(dev).eval()
imgs = [torch.rand((3, 1024, 2048), device=dev) for _ in range(2)]
lib = _lib.load()
clock_build = "CLK" in os.environ.get("SNN_HIP_LIB", "")

with torch.no_grad():
    il, _ = model.transform(imgs)
    fmap = model.backbone(il.tensors)
feats = [f.contiguous() for f in fmap.values()]
head = model.rpn.head
P = sum(f.shape[0] * f.shape[2] * f.shape[3] for f in feats)
Cw = CH // 32
lv = (_lib.snn_rpn_level * len(feats))(*[_lib.snn_rpn_level(f.data_ptr(), f.shape[0], f.shape[2], f.shape[3], 0) for f in feats])
ws_bytes = lib.snn_rpn_head_workspace_bytes(lv, len(feats), CH, A, T_RPN, head._params().precision)
stamp_off = ws_bytes // 2 + T_RPN * P * Cw * 4                   # behind the spike planes (csrc/snn_bf16x3.h, SNN_EXP_CLOCK)
tile = (C.c_int32 * 12)()
assert lib.snn_debug_tile_shape(1, P, CH, CH, T_RPN, 0, 0, tile) == 0
n_wg = int(tile[5])
ws = ops._WS.get(dev, max(ws_bytes, stamp_off + n_wg * 16 + 4096) + (200 << 20))     # one workspace for everything below (grow-only)


def read_stamps():
    if not clock_build:
        return None
    raw = ws[stamp_off: stamp_off + n_wg * 16].view(torch.int64).view(-1, 2).cpu()
    raw = raw[(raw[:, 1] > 0) & (raw[:, 0] > 0)]
    cyc, ticks = raw[:, 0].double(), raw[:, 1].double()
    return float((cyc / ticks).mean() * 0.1), float(ticks.mean() / 100.0), float(cyc.mean()), int(raw.shape[0])


def clear_stamps():
    ws[stamp_off: stamp_off + n_wg * 16].zero_()


def report(name, samples, extra=""):
    if samples and samples[0] is not None:
        ghz = statistics.mean(s[0] for s in samples)
        us = statistics.mean(s[1] for s in samples)
        cyc = statistics.mean(s[2] for s in samples)
        print("%-46s in-kernel clock %.3f GHz   K loop %.1f us / %.0f cycles per work-group (%d stamped)  %s" % (name, ghz, us, cyc, samples[0][3], extra))
    else:
        print("%-46s %s" % (name, extra))


def ev_time(fn, n):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return statistics.median(a.elapsed_time(b) for a, b in ev)


N = 20
with torch.no_grad():
    for _ in range(5):
        model(imgs)
    torch.cuda.synchronize()

    # (a) in situ
    s = []
    t0 = time.perf_counter()
    for _ in range(N):
        clear_stamps()
        model(imgs)
        torch.cuda.synchronize()
        s.append(read_stamps())
    wall = (time.perf_counter() - t0) / N * 1e3
    report("(a) in situ: model(images)", s, "wall %.2f ms per batch (with the stamp read-back)" % wall)

    # (b) heads only, as bench.py's step
    props = [torch.rand(1000, 4, device=dev) for _ in range(2)]
    for pb in props:
        pb[:, 2:] = pb[:, :2] * 600 + pb[:, 2:] * 300 + 16
        pb[:, :2] = pb[:, :2] * 600
    rois = model.roi_heads.box_roi_pool(fmap, props, il.image_sizes).contiguous() if hasattr(model.roi_heads, "box_roi_pool") else torch.randn(2000, 256, 7, 7, device=dev)
    det = model.roi_heads.box_head_and_predictor
    for _ in range(5):
        head(feats); det(rois)
    s = []
    for _ in range(N):
        clear_stamps()
        for _ in range(5):
            head(feats); det(rois)
        torch.cuda.synchronize()
        s.append(read_stamps())
    step_ms = ev_time(lambda: (head(feats), det(rois)), N)
    report("(b) heads only: RPN head + detector head", s, "step %.3f ms" % step_ms)

    # (c) conv+LIF alone, back to back
    p, w_sh = head._params(), head._packed_shared()
    head(feats)
    w_hd = head._cache_heads.val
    run_conv = lambda: ops.rpn_head_forward(feats, CH, A, T_RPN, p, w_sh, w_hd, stage_mask=2)      # noqa: E731
    ops.rpn_head_forward(feats, CH, A, T_RPN, p, w_sh, w_hd, stage_mask=7)
    s = []
    for _ in range(N):
        clear_stamps()
        for _ in range(5):
            run_conv()
        torch.cuda.synchronize()
        s.append(read_stamps())
    report("(c) conv+LIF alone, back to back", s, "launch %.4f ms (HIP events)" % ev_time(run_conv, N))

    # (c') conv+LIF alone, one launch after an idle gap
    s = []
    ms = []
    for _ in range(N):
        clear_stamps()
        time.sleep(0.02)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run_conv(); b.record()
        torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
        s.append(read_stamps())
    report("(c') conv+LIF alone, after 20 ms of idle", s, "launch %.4f ms" % statistics.median(ms))

    # (d) in situ with an idle gap in front of every forward
    s = []
    for _ in range(N):
        clear_stamps()
        time.sleep(0.02)
        model(imgs)
        torch.cuda.synchronize()
        s.append(read_stamps())
    report("(d) in situ, 20 ms idle before each forward", s)

    # (e) backbone only, then conv+LIF alone right behind it (no proposals / RoI stage in between)
    s = []
    ms = []
    for _ in range(N):
        clear_stamps()
        model.backbone(il.tensors)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run_conv(); b.record()
        torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
        s.append(read_stamps())
    report("(e) backbone, then conv+LIF alone", s, "launch %.4f ms" % statistics.median(ms))

    # (f) an L2-flushing copy (256 MB), then conv+LIF alone: cold L2 / MALL without the backbone's heat
    big_a, big_b = torch.empty(64 << 20, device=dev), torch.empty(64 << 20, device=dev)
    s = []
    ms = []
    for _ in range(N):
        clear_stamps()
        big_b.copy_(big_a)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run_conv(); b.record()
        torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
        s.append(read_stamps())
    report("(f) 256-MB copy, then conv+LIF alone", s, "launch %.4f ms" % statistics.median(ms))

# densities of the encoder's period planes e_n of this pyramid (what the conv multiplies)
os.environ["SNN_STAGE_PERIODS"] = "1"
_lib.reload_knobs()
lut = torch.tensor([bin(i).count("1") for i in range(256)], dtype=torch.int64, device=dev)
bits = torch.zeros(T_RPN, dtype=torch.float64)
n = 0
for f in feats:
    pl = ops.encode_nchw(f, T_RPN, p)
    bits += lut[pl.view(torch.uint8).to(torch.int64)].sum(dim=(1, 2)).double().cpu()
    n += f.numel()
print("period-plane densities e_1..e_%d of the in-situ pyramid: %s" % (T_RPN, " ".join("%.4f" % (b / n) for b in bits.tolist())))
