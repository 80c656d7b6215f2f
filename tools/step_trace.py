"""Kernel sequence of ONE headline step (RPN head + detector head + exchange payload) with the idle gaps between dispatches.
  on the GPU box:  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/step_trace -- python3 tools/step_trace.py run
                   python3 tools/step_trace.py report gpurun_out/step_trace"""
import csv, glob, os, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if sys.argv[1] == "run":
    import torch
    import bench
    import snn_automotive_object_detection_amd as S
    dev = torch.device("cuda:0")
    wl = dict(bench.WORKLOADS["cityscapes"])
    torch.manual_seed(4321)
    model = S.create_model(wl["dataset"], wl["K"], True, True, 0, False, False, 8, 12).to(dev).eval()
    leg = bench.Leg(wl, "bf16x3", dev, 1000, "backbone", model)
    del model
    for _ in range(5):
        leg.step()
    torch.cuda.synchronize()
    marker = torch.empty(7777, device=dev)
    marker.fill_(1.0)                                # sentinel dispatch: the steps after it are the ones reported
    for _ in range(6):
        leg.step()
    torch.cuda.synchronize()
else:
    f = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    idx = max(i for i, r in enumerate(rows) if "FillFunctor" in r["Kernel_Name"])      # the sentinel (no fill inside a step)
    rows = rows[idx + 1:]
    convs = [i for i, r in enumerate(rows) if "k_encode_levels" in r["Kernel_Name"]]
    a, b = convs[3], convs[4]                         # one step in the middle
    t_prev = int(rows[a - 1]["End_Timestamp"])
    busy = gaps = 0
    for r in rows[a:b]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print("%-64s %9.1f us   gap before %6.1f us" % (r["Kernel_Name"][:64], (e - s) / 1e3, (s - t_prev) / 1e3))
        busy += e - s
        gaps += s - t_prev
        t_prev = e
    print("step: %d dispatches, busy %.1f us, gaps %.1f us, total %.1f us" % (b - a, busy / 1e3, gaps / 1e3, (busy + gaps) / 1e3))
