import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from snn_automotive_object_detection_amd import ops as S, _lib
from tests._util import dense_to_planes
from tests.test_gpu_stages import _params
dev = torch.device("cuda:0")
os.environ["SNN_LI_HEADS"] = "mfma"; _lib.reload_knobs()
for (M, K, NA, NB) in [(300, 256, 3, 12), (300, 256, 9, 36), (300, 128, 3, 12), (300, 512, 3, 12), (77, 256, 5, 20)]:
    line = "M %d K %d NO %d:" % (M, K, NA + NB)
    for T in [2, 9, 10, 12, 18, 26]:
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
        o_a, o_b = S.li_heads(planes, K, wh, NA, NB, _params(S, "jump_first"))
        got = torch.cat([o_a, o_b], dim=1).double().cpu()
        err = (got - v).abs()
        bad = (err.amax(1) > 1e-5).nonzero().flatten().tolist()
        line += "  T%d: %.1e/%d%s" % (T, float(err.max()), len(bad), sorted(set(r % 4 for r in bad)))
    print(line)
