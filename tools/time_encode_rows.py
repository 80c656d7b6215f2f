import os, sys, torch
sys.path.insert(0, '.')
from snn_automotive_object_detection_amd import ops
dev = torch.device('cuda:0')
p = ops.make_params(ops.LIFParameters(v_th=torch.tensor(0.25)), ops.LIFParameters(alpha=100, v_th=torch.tensor(0.1)))
x = torch.randn(2000, 12544, device=dev)
def tm(fn, n=20):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return min(a.elapsed_time(b) for a, b in ev) * 1e3
for T in (12, 24):
    os.environ.pop('SNN_ENC_ROWS', None)
    print('T=%d word-per-lane %.1f us' % (T, tm(lambda: ops.encode_rows(x, T, p))))
    os.environ['SNN_ENC_ROWS'] = 'ballot'
    print('T=%d ballot        %.1f us' % (T, tm(lambda: ops.encode_rows(x, T, p))))
