"""fused linear + LIF (fc6 / fc7 shapes) across the work-group shapes of k_gemm_bf16x3 (run on the GPU box)"""
import os, sys, torch
sys.path.insert(0, '.')
from snn_automotive_object_detection_amd import ops
dev = torch.device('cuda:0')
torch.manual_seed(0)
p = ops.make_params(ops.LIFParameters(v_th=torch.tensor(0.25)), ops.LIFParameters(alpha=100, v_th=torch.tensor(0.1)))
def tm(fn, n=10):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return min(a.elapsed_time(b) for a, b in ev)
for (R, D, Hd, T) in [(2000, 12544, 1024, 12), (4000, 12544, 1024, 12), (2000, 1024, 1024, 12)]:
    x = torch.randn(R, D, device=dev)
    enc = ops.encode_rows(x, T, p)
    w = torch.randn(Hd, D, device=dev) / D ** 0.5
    wb = ops.pack_linear_bf16x3(w)
    for wn in ('2', '1'):
        for mt in ('', '4', '3', '2'):
            os.environ['SNN_BF16X3_WN'] = wn
            if mt: os.environ['SNN_BF16X3_MT'] = mt
            else: os.environ.pop('SNN_BF16X3_MT', None)
            try:
                t = tm(lambda: ops.spike_gemm_lif_bf16x3(enc, D, Hd, p, wb))
                print('R=%d D=%d  WN=%s MT=%-4s %.3f ms' % (R, D, wn, mt or 'auto', t))
            except Exception as e:
                print('R=%d D=%d  WN=%s MT=%-4s failed: %s' % (R, D, wn, mt or 'auto', str(e)[:80]))
os.environ.pop('SNN_BF16X3_WN', None); os.environ.pop('SNN_BF16X3_MT', None)
