"""Where the RPN post-processing time goes (stage timings with syncs; run on the GPU box)."""
import sys, time, torch
sys.path.insert(0, '.')
import snn_automotive_object_detection_amd as S
from snn_automotive_object_detection_amd.rpn import concat_box_prediction_layers
from snn_automotive_object_detection_amd.stock import boxes as box_ops
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = S.create_model('cityscapes', 9, True, True, 0, False, False, 8, 12).to(dev).eval()
imgs = [torch.rand((3, 1024, 2048), device=dev) for _ in range(2)]
def sync(): torch.cuda.synchronize()
def T(label, fn, n=5):
    fn(); sync(); t0 = time.perf_counter()
    for _ in range(n): r = fn()
    sync(); print('%-28s %.3f ms' % (label, (time.perf_counter() - t0) * 1e3 / n)); return r
with torch.no_grad():
    il, _ = m.transform(imgs)
    feats = m.backbone(il.tensors)
    fl = list(feats.values())
    rpn = m.rpn
    obj, dl = T('head', lambda: rpn.head(fl)[:2])
    anchors = T('anchors', lambda: rpn.anchor_generator(il, fl))
    napl = [o.shape[1] * o.shape[2] * o.shape[3] for o in obj]
    o2, d2 = T('concat', lambda: concat_box_prediction_layers(obj, dl))
    props = T('decode', lambda: rpn.box_coder.decode(d2.detach(), anchors).view(len(anchors), -1, 4))
    T('filter (reference order)', lambda: rpn.filter_proposals_reference(props, o2, il.image_sizes, napl))
    T('filter (batched)', lambda: rpn.filter_proposals(o2, d2, anchors, il.image_sizes, napl))
    ob = o2.detach().reshape(2, -1)
    T('topk x5', lambda: [x.topk(min(1000, x.shape[1]), dim=1)[1] for x in ob.split(napl, 1)])
    T('rpn forward (all)', lambda: rpn(il, feats))
