import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import snn_automotive_object_detection_amd as S
from snn_automotive_object_detection_amd import _lib
from oracle import snn_oracle as OR
dev = torch.device("cuda:0")
for T in [int(x) for x in sys.argv[1:]] or [10]:
    torch.manual_seed(T)
    m = S.RPNHeadSNN(256, 3, T).to(dev)
    with torch.no_grad():
        m.shared_conv.weight.mul_(4.0)
    g = torch.Generator().manual_seed(T)
    feats = [(torch.randn(2, 256, h, w, generator=g) * 1.7) for h, w in [(37, 53), (19, 27), (7, 9), (1, 3)]]
    fd = [f.to(dev) for f in feats]
    os.environ.pop("SNN_SPARSE", None); _lib.reload_knobs()
    lg, bb = m(fd); a = [t.cpu() for t in lg + bb]; pa = _lib.load().snn_debug_last_conv_path()
    os.environ["SNN_SPARSE"] = "0"; _lib.reload_knobs()
    lg, bb = m(fd); b = [t.cpu() for t in lg + bb]; pb_ = _lib.load().snn_debug_last_conv_path()
    o_l, o_b = OR.rpn_head_forward(feats, m.shared_conv.weight.detach().cpu(), m.conv_cls.weight.detach().cpu(), m.conv_bbox.weight.detach().cpu(), T)
    o = list(o_l) + list(o_b)
    L = len(feats)
    def off(x, y):
        res = []
        base = 0
        for l in range(L):
            d = torch.maximum((x[l] - y[l]).abs().amax(1), (x[L + l] - y[L + l]).abs().amax(1))
            idx = (d > 1e-4).nonzero()
            N, H, W = d.shape
            for n, yy, xx in idx.tolist():
                res.append((l, n, yy, xx, base + (n * H + yy) * W + xx, round(float(d[n, yy, xx]), 5)))
            base += N * H * W
        return res
    print("T", T, "paths", pa, pb_)
    for name, r in [("sparse vs oracle", off(a, o)), ("dense vs oracle", off(b, o)), ("sparse vs dense", off(a, b))]:
        print(" ", name, len(r), r[:24])
