"""Backbone (stock MIOpen fp32 ResNet-50-FPN) timing under the settings stock PyTorch offers: memory format and
MIOpen's find mode.  Not our kernels; decides what create_model() should switch on for the end-to-end leg."""
import sys, time, torch
sys.path.insert(0, '.')
import snn_automotive_object_detection_amd as S
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = S.create_model('cityscapes', 9, True, True, 0, False, False, 8, 12).to(dev).eval()
imgs = [torch.rand((3, 1024, 2048), device=dev) for _ in range(2)]
def sync(): torch.cuda.synchronize()
def timed(fn, n=10):
    for _ in range(3): fn()
    sync(); t0 = time.perf_counter()
    for _ in range(n): fn()
    sync(); return (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
    il, _ = m.transform(imgs)
    x = il.tensors
    ref = m.backbone(x)
    for bench in (False, True):
        torch.backends.cudnn.benchmark = bench
        for cl in (False, True):
            bb = m.backbone.to(memory_format=torch.channels_last) if cl else m.backbone.to(memory_format=torch.contiguous_format)
            xi = x.contiguous(memory_format=torch.channels_last) if cl else x.contiguous()
            out = bb(xi)
            err = max(float((out[k] - ref[k]).abs().max()) for k in ref)
            strides = out['0'].stride()
            ms = timed(lambda: bb(xi))
            ms_c = timed(lambda: {k: v.contiguous() for k, v in bb(xi).items()})
            print('benchmark=%s channels_last=%s: backbone %.2f ms (+contiguous outputs %.2f ms) max|diff| %.2e out stride %s' % (bench, cl, ms, ms_c, err, strides), flush=True)
