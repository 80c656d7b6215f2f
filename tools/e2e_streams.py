"""End-to-end throughput with S batches in flight, one Python thread + one HIP stream each (the forward has host
syncs - per-image proposal / detection counts - so one thread cannot keep two streams fed)."""
import sys, time, threading, torch
sys.path.insert(0, '.')
import snn_automotive_object_detection_amd as S
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = S.create_model('cityscapes', 9, True, True, 0, False, False, 8, 12).to(dev).eval()
imgs = [torch.rand((3, 1024, 2048), device=dev) for _ in range(2)]
with torch.no_grad():
    for _ in range(3): out = m(imgs)
torch.cuda.synchronize()
ref = [d['boxes'].shape[0] for d in out]

def run(n_streams, n_batches=24):
    streams = [torch.cuda.Stream(dev) for _ in range(n_streams)]
    res = [None] * n_streams
    def worker(i):
        with torch.no_grad(), torch.cuda.stream(streams[i]):
            for _ in range(n_batches // n_streams):
                o = m(imgs)
            res[i] = [d['boxes'].shape[0] for d in o]
            streams[i].synchronize()
    for i in range(n_streams):                      # warm every stream's workspaces / allocator pools
        with torch.no_grad(), torch.cuda.stream(streams[i]): m(imgs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(i,)) for i in range(n_streams)]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nb = (n_batches // n_streams) * n_streams
    print('%d stream(s): %.2f ms per batch, %.1f img/s; detections %s (ref %s)' % (n_streams, dt / nb * 1e3, 2 * nb / dt, res, ref), flush=True)

for s in (1, 2, 3, 1, 2):
    run(s)
