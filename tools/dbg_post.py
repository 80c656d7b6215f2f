"""debug: where do the HIP post-processing results leave the oracle's (run on the GPU box)"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from oracle import fixtures as FX, post_oracle as PO, torchvision_restated as TV
from tests.test_post_golden import product_rpn
import snn_automotive_object_detection_amd as S

dev = torch.device("cuda:0")
np.set_printoptions(precision=6, suppress=True, linewidth=200)


def first_div(g, e, tol=1e-3):
    n = min(len(g), len(e))
    for i in range(n):
        if np.abs(g[i] - e[i]).max() > tol:
            return i
    return n if len(g) != len(e) else -1


# (a) RPN fixture with the threshold edge
sp = FX.RPN_POST_SPECS["post_rpn_empty"]
exp = FX.load_expected("post_rpn_empty")
rpn, images, feats = product_rpn(sp, dev)
boxes, pre = rpn(images, feats)
eb = exp["boxes"].reshape(-1, 4)[int(exp["boxes_n"][0]):]
g = boxes[1].cpu().numpy()
i = first_div(g, eb)
print("rpn_empty image 1: got", g.shape, "exp", eb.shape, "first divergence row", i)
print("got", g[max(0, i - 1):i + 3]); print("exp", eb[max(0, i - 1):i + 3])
pp = pre[1]["objectness"].cpu().numpy(); ep = exp["pre_prob"][1]
print("candidates >= thr: hip", int((pp >= np.float32(0.9999)).sum()), "ref", int((ep >= np.float32(0.9999)).sum()),
      "near thr (1e-5):", int((np.abs(ep - 0.9999) < 1e-5).sum()))
so, se = np.sort(pp)[::-1], np.sort(ep)[::-1]
print("max |prob diff| sorted", np.abs(so - se).max())

# (b) detector full size
sp = dict(K=9, rois=[1000, 1000], image_shapes=[(768, 1536), (750, 1500)], seed=451, logit_std=2.5, delta_std=0.8, clusters=40)
logits, reg, props = FX.det_post_inputs(sp)
st = {}
e = PO.det_postprocess(logits, reg, props, list(sp["image_shapes"]), stats=st)
heads = S.RoIHeadsSNN(None, None, 0.5, 0.5, 512, 0.25, None, 0.4, 0.5, 100)
r = heads.postprocess_detections(logits.to(dev), reg.to(dev), [p.to(dev) for p in props], list(sp["image_shapes"]))
for i in range(2):
    gl, el = r[2][i].cpu().numpy(), e[2][i].numpy()
    gb, eb_ = r[0][i].cpu().numpy()[gl == 0], e[0][i].numpy()[el == 0]
    gs, es = r[1][i].cpu().numpy()[gl == 0], e[1][i].numpy()[el == 0]
    j = first_div(gb, eb_)
    print("det image", i, "bg rows hip", gb.shape[0], "oracle", eb_.shape[0], "first divergence", j)
    if j >= 0:
        print("hip", gb[max(0, j - 1):j + 3], gs[max(0, j - 1):j + 3]); print("ora", eb_[max(0, j - 1):j + 3], es[max(0, j - 1):j + 3])
    # which RoIs count as background on either side (before NMS)?
    sc = torch.softmax(logits[i * 1000:(i + 1) * 1000], -1)
    n_bg_ref = int(((sc[:, 1:] > 0.4).sum(1) == 0).sum())
    asg = r[3][i].cpu()
    n_bg_hip = int(((asg[:, 1:] > 0.4).sum(1) == 0).sum())
    print("   RoIs without a foreground class: ref", n_bg_ref, "hip scores", n_bg_hip, " max |softmax diff|", float((asg - sc).abs().max()))
    # NMS of the oracle on the HIP side's own bg candidates
    bgm = ((asg[:, 1:] > 0.4).sum(1) == 0)
    bb = r[4][i].cpu()[bgm][:, 0]
    ss = asg[bgm][:, 0]
    keep0 = TV.remove_small_boxes(bb, 1e-2)
    k = TV.nms(bb[keep0], ss[keep0], 0.5)
    print("   oracle NMS on HIP's own bg boxes keeps", len(k))

# (c) random-init head outputs: exact ties
torch.manual_seed(0)
m = S.create_model("cityscapes", 9, True, True, 0, False, False, num_steps_rpn=4, num_steps_detector=4)
m.transform.min_size, m.transform.max_size = 256, 512
m = m.to(dev).eval()
gg = torch.Generator().manual_seed(3)
imgs = [torch.rand((3, 512, 1024), generator=gg).to(dev) for _ in range(3)]
with torch.no_grad():
    il, _ = m.transform(imgs)
    fm = m.backbone(il.tensors)
    lo, de = m.rpn.head(list(fm.values()))
for l, o in enumerate(lo):
    v = o.flatten().cpu().numpy()
    u = np.unique(v)
    print("level", l, "logits", v.size, "distinct", u.size, "zeros", int((v == 0).sum()), "range", v.min(), v.max())
