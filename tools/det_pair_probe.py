"""fc6 + fc7 of the detector head, two launches against one (SNN_DET_PAIR=1), on the bench's own RoI features:
  python tools/det_pair_probe.py run  N          -> runs the detector head N times (under rocprofv3: kernel trace / PMC)
  python tools/det_pair_probe.py report DIR...   -> per-dispatch durations and counters of k_gemm_bf16x3 / k_gemm_bf16x3_pair, fc6 and fc7
                                                    told apart by dispatch order (fc6, fc7, fc6, ...)"""
import csv
import glob
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(n):
    import torch
    import bench
    import snn_automotive_object_detection_amd as S
    dev = torch.device("cuda:0")
    wl = dict(bench.WORKLOADS["cityscapes"])
    torch.manual_seed(4321)
    model = S.create_model(wl["dataset"], wl["K"], True, True, 0, False, False, 8, 12).to(dev).eval()
    leg = bench.Leg(wl, "bf16x3", dev, 1000, "backbone", model)
    del model
    for _ in range(n):
        leg.det_head(leg.rois)
    torch.cuda.synchronize()


def report(dirs):
    for d in dirs:
        print("==", d)
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            rows = [r for r in csv.DictReader(open(f)) if "k_gemm_bf16x3" in r["Kernel_Name"]]
            rows.sort(key=lambda r: int(r["Start_Timestamp"]))
            dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
            if any("pair" in r["Kernel_Name"] for r in rows):
                print("  pair launch: n=%d  median %.1f us  min %.1f us" % (len(dur), statistics.median(dur), min(dur)))
            else:
                a, b = dur[0::2], dur[1::2]
                gaps = [(int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"])) / 1e3 for i in range(0, len(rows) - 1, 2)]
                print("  fc6: n=%d median %.1f us min %.1f | fc7: n=%d median %.1f us min %.1f | gap fc6 -> fc7 median %.1f us | fc6 + gap + fc7 median %.1f us" % (
                    len(a), statistics.median(a), min(a), len(b), statistics.median(b), min(b), statistics.median(gaps),
                    statistics.median([x + y + g for x, y, g in zip(a, b, gaps)])))
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per = {}
            for r in csv.DictReader(open(f)):
                if "k_gemm_bf16x3" in r["Kernel_Name"]:
                    per.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"]})[r["Counter_Name"]] = float(r["Counter_Value"])
            ids = sorted(per)
            groups = {"pair": ids} if any("pair" in per[i]["name"] for i in ids) else {"fc6": ids[0::2], "fc7": ids[1::2]}
            for g, sel in groups.items():
                names = sorted(k for k in per[sel[0]] if k != "name")
                mean = {k: statistics.mean(per[i][k] for i in sel) for k in names}
                line = "  %-4s n=%d  " % (g, len(sel)) + "  ".join("%s=%.4g" % (k, v) for k, v in mean.items())
                if "SQ_VALU_MFMA_BUSY_CYCLES" in mean and "GRBM_GUI_ACTIVE" in mean:
                    line += "  -> matrix pipe busy %.1f %%" % (100.0 * mean["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * mean["GRBM_GUI_ACTIVE"] / 8))
                print(line)


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]))
    else:
        report(sys.argv[2:])
