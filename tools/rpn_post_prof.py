"""rocprofv3 target: 20 x RegionProposalNetwork.filter_proposals on fixed head outputs."""
import sys, torch
sys.path.insert(0, '.')
import snn_automotive_object_detection_amd as S
from snn_automotive_object_detection_amd.rpn import concat_box_prediction_layers
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = S.create_model('cityscapes', 9, True, True, 0, False, False, 8, 12).to(dev).eval()
imgs = [torch.rand((3, 1024, 2048), device=dev) for _ in range(2)]
with torch.no_grad():
    il, _ = m.transform(imgs)
    feats = m.backbone(il.tensors)
    fl = list(feats.values())
    rpn = m.rpn
    obj, dl = rpn.head(fl)[:2]
    anchors = rpn.anchor_generator(il, fl)
    napl = [o.shape[1] * o.shape[2] * o.shape[3] for o in obj]
    o2, d2 = concat_box_prediction_layers(obj, dl)
    torch.cuda.synchronize()
    for _ in range(20):
        rpn.filter_proposals(o2, d2, anchors, il.image_sizes, napl)
    torch.cuda.synchronize()
