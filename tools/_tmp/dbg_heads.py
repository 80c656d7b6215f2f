import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from snn_automotive_object_detection_amd import ops as S, _lib
from tests._util import dense_to_planes
from tests.test_gpu_stages import _params
dev = torch.device("cuda:0")
M, K, NA, NB = 300, 256, 3, 12
for T in [8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18]:
    g = torch.Generator().manual_seed(T)
    spk = torch.rand(T, M, K, generator=g) < 0.1
    wa = torch.randn(NA, K, generator=g) / K ** 0.5
    wb = torch.randn(NB, K, generator=g) / K ** 0.5
    a, b = 0.1, 0.2
    cur = torch.einsum("tmk,nk->tmn", spk.double(), torch.cat([wa, wb]).double())
    v = torch.zeros(M, NA + NB, dtype=torch.float64); i = torch.zeros_like(v)
    for t in range(T):
        i = i + cur[t]; v = v + a * (i - v); i = i - b * i
    planes = dense_to_planes(spk.numpy()).to(dev)
    wh = S.pack_heads(wa.to(dev), wb.to(dev))
    line = "T %2d" % T
    for form in ("valu", "ksplit", "mfma"):
        os.environ["SNN_LI_HEADS"] = form; _lib.reload_knobs()
        o_a, o_b = S.li_heads(planes, K, wh, NA, NB, _params(S, "jump_first"))
        got = torch.cat([o_a, o_b], dim=1).double().cpu()
        err = (got - v).abs().amax(1)
        bad = (err > 1e-5).nonzero().flatten().tolist()
        line += "  %s: max %.2e bad rows %d %s" % (form, float(err.max()), len(bad), sorted(set(r % 4 for r in bad)))
    print(line)
