"""Is the stock backbone (MIOpen convolutions) bitwise repeatable run to run, and does any switch make it so?  (run on the GPU box)
Prints, per setting, whether two forward passes of the ResNet-50-FPN on the same 2 x 720x1280 batch give identical pyramids."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import snn_automotive_object_detection_amd as S

dev = torch.device("cuda:0")
torch.manual_seed(4321)
m = S.create_model("bdd", 11, True, True, 0, False, False, 8, 12).to(dev).eval()
g = torch.Generator().manual_seed(1)
imgs = [torch.rand((3, 720, 1280), generator=g).to(dev) for _ in range(2)]


def run():
    with torch.no_grad():
        il, _ = m.transform(imgs)
        return [v.clone() for v in m.backbone(il.tensors).values()]


def same(a, b):
    return [bool(torch.equal(x, y)) for x, y in zip(a, b)], max(float((x - y).abs().max()) for x, y in zip(a, b))


for name, setup in (("default", lambda: None),
                    ("cudnn.deterministic", lambda: setattr(torch.backends.cudnn, "deterministic", True)),
                    ("use_deterministic_algorithms", lambda: torch.use_deterministic_algorithms(True, warn_only=True)),
                    ("cudnn.benchmark", lambda: setattr(torch.backends.cudnn, "benchmark", True))):
    setup()
    run()
    a, b, c = run(), run(), run()
    print("%-32s a==b %s  a==c %s" % (name, same(a, b), same(a, c)), flush=True)
    det = [m(imgs) for _ in range(2)]
    print("%-32s detections equal: %s (counts %s / %s)" % ("", all(torch.equal(x["boxes"], y["boxes"]) for x, y in zip(*det)),
                                                        [int(d["boxes"].shape[0]) for d in det[0]], [int(d["boxes"].shape[0]) for d in det[1]]), flush=True)
