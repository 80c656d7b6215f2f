"""Per-step GPU time inside bench.py's timed window (HIP events around every step): is the driver's 20-step window slower per step than the 100-step one because its first steps are?
   python tools/step_times.py [steps] [warmup]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                                     # noqa: E402
import bench                                                     # noqa: E402
import snn_automotive_object_detection_amd as S                  # noqa: E402

steps, warmup = int(sys.argv[1]) if len(sys.argv) > 1 else 20, int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
wl = dict(bench.WORKLOADS["cityscapes"])
torch.manual_seed(4321)
model = S.create_model(wl["dataset"], wl["K"], True, True, 0, False, False, 8, 12).to(dev).eval()
leg = bench.Leg(wl, "bf16x3", dev, 1000, "backbone", model)
del model
for rep in range(3):
    for _ in range(warmup):
        leg.step()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(steps):
        leg.step()
        ev[i + 1].record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]
    print("window %d: wall %.3f ms = %.4f ms/step; GPU span first event -> last %.3f ms; per step: %s" % (
        rep, wall, wall / steps, ev[0].elapsed_time(ev[steps]), " ".join("%.3f" % m for m in ms)))
    time.sleep(0.5)
