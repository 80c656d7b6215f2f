"""Debug helper: one bf16x3 spike GEMM of a given shape against a float64 product of the same operands."""
import sys, torch
sys.path.insert(0, '.')
from snn_automotive_object_detection_amd import ops
torch.manual_seed(0)
M, K, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
Kw = (K + 31) // 32
bits = (torch.rand(M, Kw * 32, device='cuda') < 0.3)
bits[:, K:] = False
wts = (1 << torch.arange(32, device='cuda', dtype=torch.int64))
a = (bits.view(M, Kw, 32).to(torch.int64) * wts).sum(-1)
a = torch.where(a >= 2**31, a - 2**32, a).to(torch.int32)
w = torch.randn(N, K, device='cuda')
wp = ops.pack_linear_bf16x3(w)
cur = ops.spike_gemm_bf16x3(a, K, N, wp)
torch.cuda.synchronize()
ref = bits[:, :K].double() @ w.double().t()
print('ok', tuple(cur.shape), 'max abs err', float((cur.double() - ref).abs().max()))
if len(sys.argv) > 4:
    Kc = Kw
    parts = [bits[:, 32 * c:32 * c + 32].double() @ w.double().t()[32 * c:32 * c + 32] for c in range(Kc)]
    import itertools
    for sel in itertools.product([0, 1, 2], repeat=Kc):
        cand = sum(s * p for s, p in zip(sel, parts))
        e = float((cur.double() - cand).abs().max())
        if e < 1e-3:
            print('matches chunk multiplicities', sel)
