"""Period planes e_n = (first spike at step n-1) of the bench's REAL encoder spike trains next to the spike planes z_t themselves:
density per plane and the time of the stage-level conv+LIF / fc6+LIF launches on either (timing only: on e_n planes the launch
computes u_n = W e_n, which a period-aware epilogue would recombine into the currents).  Run on the GPU box."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import snn_automotive_object_detection_amd as S
from snn_automotive_object_detection_amd import ops

dev = torch.device("cuda:0")
wl = dict(bench.WORKLOADS["cityscapes"])
torch.manual_seed(4321)
model = S.create_model(wl["dataset"], wl["K"], True, True, 0, False, False, 8, 12).to(dev).eval()
leg = bench.Leg(wl, "bf16x3", dev, 1000, "backbone", model)
del model
p = leg.rpn_head._params()
shapes = [(int(f.shape[0]), int(f.shape[2]), int(f.shape[3])) for f in leg.feats]
lut = torch.tensor([bin(i).count("1") for i in range(256)], dtype=torch.int64, device=dev)


def dens(pl):
    return [round(float(lut[x.contiguous().view(torch.uint8).to(torch.int64)].sum()) / (x.numel() * 32), 4) for x in pl]


def periods(z):
    """z [T, rows, words] -> e [T, rows, words]: e[n-1] = bits whose FIRST spike is at step n-1"""
    seen = torch.zeros_like(z[0])
    out = []
    for t in range(z.shape[0]):
        out.append(z[t] & ~seen)
        seen = seen | z[t]
    return torch.stack(out)


def tm(fn, n=15):
    fn(); torch.cuda.synchronize()
    best = []
    for _ in range(3):
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for a, b in ev:
            a.record(); fn(); b.record()
        torch.cuda.synchronize()
        best.append(sum(a.elapsed_time(b) for a, b in ev) / n)
    return min(best)


T = 8
z = torch.cat([ops.encode_nchw(f, T, p) for f in leg.feats], dim=1).contiguous()
e = periods(z)
print("RPN  z_t density", dens(z), " e_n density", dens(e))
wb = leg.rpn_head._packed_shared()
zp, ep = ops.pad_planes(z, shapes), ops.pad_planes(e, shapes)
for rnd in range(2):
    print("conv+LIF on z planes %.4f ms   on e planes %.4f ms" % (tm(lambda: ops.conv3x3_lif_bf16x3(zp, shapes, 256, 256, p, wb)),
                                                                 tm(lambda: ops.conv3x3_lif_bf16x3(ep, shapes, 256, 256, p, wb))), flush=True)
Td = 12
pd = leg.det_head._params()
zd = ops.encode_rows(leg.rois.flatten(1), Td, pd)
ed = periods(zd)
print("DET  z_t density", dens(zd), " e_n density", dens(ed))
w6 = leg.det_head._packed(inner=0)[0]
for rnd in range(2):
    print("fc6+LIF on z planes %.4f ms   on e planes %.4f ms" % (tm(lambda: ops.spike_gemm_lif_bf16x3(zd, 12544, 1024, pd, w6)),
                                                                tm(lambda: ops.spike_gemm_lif_bf16x3(ed, 12544, 1024, pd, w6))), flush=True)
