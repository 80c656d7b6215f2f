"""bench.Leg.kernel_breakdown of the headline workload (RPN encoder / conv+LIF / RPN head / detector head), N rounds (run on the GPU box)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import snn_automotive_object_detection_amd as S

dev = torch.device("cuda:0")
wl = dict(bench.WORKLOADS[os.environ.get("AB_WORKLOAD", "cityscapes")])
torch.manual_seed(4321)
model = S.create_model(wl["dataset"], wl["K"], True, True, 0, False, False, 8, 12).to(dev).eval()
leg = bench.Leg(wl, "bf16x3", dev, 1000, "backbone", model)
del model
leg.step_local()                                               # packs the weights
for _ in range(int(os.environ.get("AB_ROUNDS", "3"))):
    print({k: round(v, 4) for k, v in leg.kernel_breakdown(20).items()}, flush=True)
