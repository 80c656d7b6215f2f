"""Per-work-group timeline of the conv+LIF launch (a -DSNN_EXP_TIMELINE build: s_memrealtime stamps at kernel entry, K-loop start,
K-loop end, exit, + HW_ID / XCC_ID): how long a work-group spends before / in / after its K loop, and how long a CU's slot stays
empty between one work-group's exit and the next one's entry.
   bash tools/ab_build.sh TL:"-DSNN_EXP_TIMELINE"  (here), then on the GPU box:  python tools/wg_timeline.py tools/_ab/lib_TL.so"""
import ctypes as C
import os
import sys
from collections import defaultdict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snn_automotive_object_detection_amd import _lib, ops

os.environ["SNN_STAGE_PERIODS"] = "1"
os.environ["SNN_STAGE_PLANES"] = "wm"
_lib.reload_knobs()
lib = C.CDLL(sys.argv[1])
lib.snn_debug_reload_knobs()
for n, (res, at) in (list(_lib.SYMBOLS.items()) + list(_lib.DEBUG_SYMBOLS.items())):
    if hasattr(lib, n):
        getattr(lib, n).restype = res
        getattr(lib, n).argtypes = at
dev = torch.device("cuda:0")
torch.manual_seed(0)
p = ops.make_params(ops.LIFParameters(v_th=torch.tensor(0.25)), ops.LIFParameters(alpha=100, v_th=torch.tensor(0.1)))
LEVELS = [(192, 384), (96, 192), (48, 96), (24, 48), (12, 24)]
T = 8
os.environ["SNN_STAGE_PLANES"] = "rm"; _lib.reload_knobs()
feats = [torch.randn(2, 256, h, w, device=dev) * 1.7 for h, w in LEVELS]
encs = torch.cat([ops.encode_nchw(f, T, p) for f in feats], dim=1).contiguous()              # period planes (SNN_STAGE_PERIODS=1)
encs_pad = ops.pad_planes(encs, [(2, h, w) for h, w in LEVELS])
enc_wm = encs_pad.permute(0, 2, 1).contiguous()
PP, P = encs_pad.shape[1], encs.shape[1]
wb = ops.pack_conv3x3_bf16x3(torch.randn(256, 256, 3, 3, device=dev) * 0.01)
lv = (_lib.snn_rpn_level * 5)(*[_lib.snn_rpn_level(None, 2, h, w_, 0) for h, w_ in LEVELS])
N_WG = 16384
buf = torch.zeros((T * P * 8 + N_WG * 32,), dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream


def run():
    rc = lib.snn_conv3x3_lif_bf16x3(enc_wm.data_ptr(), PP * 8, lv, 5, 256, 256, T, C.byref(p), wb.data_ptr(), buf.data_ptr(), P * 8, st)
    assert rc == 0, lib.snn_last_error()


if len(sys.argv) > 2 and sys.argv[2] == "fc6":                      # the detector's fc6 + LIF launch instead (2000 RoIs, T = 12)
    R, D, Hd, T6 = 2000, 12544, 1024, 12
    x = torch.randn(R, D, device=dev) * 1.7
    enc6 = ops.encode_rows(x, T6, p).permute(0, 2, 1).contiguous()  # period planes, word-major
    w6b = ops.pack_linear_bf16x3(torch.randn(Hd, D, device=dev) / D ** 0.5)
    T, P = T6, R * (Hd // 32) // 8                                   # (so that T * P * 8 = the spike planes' size)
    buf = torch.zeros((T6 * R * (Hd // 32) + N_WG * 32,), dtype=torch.int32, device=dev)

    def run():
        rc = lib.snn_spike_gemm_lif_bf16x3(enc6.data_ptr(), T6, R, D, Hd, C.byref(p), w6b.data_ptr(), buf.data_ptr(), R * (Hd // 32), st)
        assert rc == 0, lib.snn_last_error()

for _ in range(10):
    run()
torch.cuda.synchronize()
buf[T * P * 8:].zero_()
run()
torch.cuda.synchronize()
tl = buf[T * P * 8:].view(torch.int64).view(-1, 16).cpu()
tl = tl[tl[:, 0] > 0]
print("work-groups stamped:", len(tl))
t0 = int(tl[:, 0].min())
ent, l0, l1, ex = [(tl[:, i] - t0).double() * 0.01 for i in range(4)]           # microseconds (100 MHz)
print("kernel span %.1f us" % float(ex.max()))
print("per work-group (us): before the K loop %.2f   K loop %.2f   after it (LIF epilogue) %.2f   total %.2f" %
      (float((l0 - ent).mean()), float((l1 - l0).mean()), float((ex - l1).mean()), float((ex - ent).mean())))
for name, v in (("before", l0 - ent), ("loop", l1 - l0), ("after", ex - l1)):
    q = torch.quantile(v, torch.tensor([0.05, 0.5, 0.95], dtype=torch.float64))
    print("   %-6s p5 %.2f  median %.2f  p95 %.2f" % (name, *[float(x) for x in q]))
ph = [(tl[:, 8 + i] - t0).double() * 0.01 for i in range(7)]
names = ["K loop end -> DMA drained", "-> barrier 1 (pass 0)", "-> accumulators in LDS + barrier 2", "-> LIF + stores of pass 0 (wave 0)",
         "-> barrier 1 (pass 1)", "-> accumulators in LDS + barrier 2", "-> LIF + stores of pass 1 (wave 0)"]
prev = l1
for n_, v in zip(names, ph):
    print("   epilogue %-46s %.2f us" % (n_, float((v - prev).mean())))
    prev = v
print("   epilogue %-46s %.2f us" % ("-> exit (wave 0)", float((ex - prev).mean())))
# slots: (xcc, se, sh, cu) -> its work-groups in time order; gap = next entry - previous exit (two slots per CU interleave:
# pair each entry with the latest exit on that CU before it)
hw, xcc = tl[:, 4], tl[:, 5] & 0xF
cu_key = (xcc * 4096 + ((hw >> 13) & 0x7) * 256 + ((hw >> 12) & 1) * 64 + ((hw >> 8) & 0xF)).tolist()
by_cu = defaultdict(list)
for k, a, b in zip(cu_key, ent.tolist(), ex.tolist()):
    by_cu[k].append((a, b))
print("CUs seen:", len(by_cu), " work-groups per CU: min %d max %d" % (min(len(v) for v in by_cu.values()), max(len(v) for v in by_cu.values())))
gaps, idle, both = [], 0.0, 0.0
for k, v in by_cu.items():
    exits = sorted(b for _, b in v)
    entries = sorted(a for a, _ in v)
    # greedy: every entry after the first two takes over the slot freed by the earliest exit not yet reused
    for i, a in enumerate(entries[2:]):
        gaps.append(a - exits[i])
g = torch.tensor(gaps, dtype=torch.float64)
print("slot turnover (entry of the next work-group - exit of the one it replaces): mean %.2f us  median %.2f  p95 %.2f  (n=%d)" %
      (float(g.mean()), float(g.median()), float(torch.quantile(g, 0.95)), len(g)))
