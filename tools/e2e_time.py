import sys, time, torch
sys.path.insert(0, '.')
import snn_automotive_object_detection_amd as S
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = S.create_model('cityscapes', 9, True, True, 0, False, False, 8, 12).to(dev).eval()
imgs = [torch.rand((3, 1024, 2048), device=dev) for _ in range(2)]
def sync(): torch.cuda.synchronize()
for it in range(3):
    sync(); t0 = time.perf_counter(); out = m(imgs); sync(); print('e2e iter', it, (time.perf_counter() - t0) * 1e3, 'ms', [d['boxes'].shape[0] for d in out])
# stage timing
from snn_automotive_object_detection_amd.stock.anchors import ImageList
with torch.no_grad():
    sync(); t0 = time.perf_counter(); il, _ = m.transform(imgs); sync(); t1 = time.perf_counter()
    feats = m.backbone(il.tensors); sync(); t2 = time.perf_counter()
    props, extra = m.rpn(il, feats); sync(); t3 = time.perf_counter()
    bf = m.roi_heads.box_roi_pool(feats, props, il.image_sizes); sync(); t4 = time.perf_counter()
    cl, bx = m.roi_heads.box_head_and_predictor(bf); sync(); t5 = time.perf_counter()
    res = m.roi_heads.postprocess_detections(cl, bx, props, il.image_sizes); sync(); t6 = time.perf_counter()
print('transform %.1f backbone %.1f rpn(all) %.1f roialign %.1f dethead %.1f postprocess %.1f ms' % tuple(1e3 * x for x in (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5)), 'R=', bf.shape[0])
