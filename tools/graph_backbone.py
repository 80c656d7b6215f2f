"""Does the stock backbone capture into a hipGraph (torch.cuda.CUDAGraph), and what does a replay cost against eager?"""
import sys, time, torch
sys.path.insert(0, '.')
import snn_automotive_object_detection_amd as S
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = S.create_model('cityscapes', 9, True, True, 0, False, False, 8, 12).to(dev).eval()
imgs = [torch.rand((3, 1024, 2048), device=dev) for _ in range(2)]
sync = torch.cuda.synchronize
def timed(fn, n=20):
    for _ in range(3): fn()
    sync(); t0 = time.perf_counter()
    for _ in range(n): fn()
    sync(); return (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
    il, _ = m.transform(imgs)
    x = il.tensors.clone()
    print('eager backbone %.2f ms' % timed(lambda: m.backbone(x)), flush=True)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): m.backbone(x)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = m.backbone(x)
    sync()
    print('captured', flush=True)
    ref = m.backbone(x)
    g.replay(); sync()
    print('max diff replay vs eager', max(float((out[k] - ref[k]).abs().max()) for k in ref), flush=True)
    print('graph replay %.2f ms' % timed(lambda: g.replay()), flush=True)
