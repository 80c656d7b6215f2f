"""CPU checks of the post-processing pins.  tests/golden/post_*.npz were written by
oracle/make_golden.py from the reference's OWN RegionProposalNetwork.forward (rpn.py:563-703 incl. filter_proposals 420-499)
and RoIHeadsSNN.postprocess_detections (roi_heads.py:1075-1176).  Here: (1) the oracle restatement (oracle/post_oracle.py)
reproduces them bit for bit; (2) the product's stock-torch CPU comparators (`post="reference"`) agree with them."""
import numpy as np
import pytest
import torch

from oracle import fixtures as FX
from oracle import post_oracle as PO
from tests._util import assert_same_detections, split_rows


class StoredHead(torch.nn.Module):
    def __init__(self, obj, dl):
        super().__init__()
        self.obj, self.dl = obj, dl

    def forward(self, feats):
        return self.obj, self.dl


@pytest.mark.parametrize("name", sorted(FX.RPN_POST_SPECS))
def test_rpn_post_oracle_equals_reference_fixture(name):
    sp, exp = FX.RPN_POST_SPECS[name], FX.load_expected(name)
    obj, dl = FX.rpn_post_inputs(sp)
    b, s, pre = PO.rpn_proposals(obj, dl, sp["canvas"], sp["image_sizes"], FX.ANCHOR_SIZES, FX.ASPECT_RATIOS, sp["pre"], sp["post"],
                                 sp["nms"], sp["score_thresh"])
    assert [int(x.shape[0]) for x in b] == exp["boxes_n"].tolist()
    assert np.array_equal(torch.cat(b).numpy().reshape(-1, 4), exp["boxes"].reshape(-1, 4))
    assert np.array_equal(torch.cat(s).numpy().reshape(-1), exp["scores"].reshape(-1))
    assert np.array_equal(torch.stack([p["proposals"] for p in pre]).numpy(), exp["pre_boxes"])
    assert np.array_equal(torch.stack([p["objectness"] for p in pre]).numpy(), exp["pre_prob"])


@pytest.mark.parametrize("name", sorted(FX.DET_POST_SPECS))
def test_det_post_oracle_equals_reference_fixture(name):
    sp, exp = FX.DET_POST_SPECS[name], FX.load_expected(name)
    logits, reg, props = FX.det_post_inputs(sp)
    out = PO.det_postprocess(logits, reg, props, list(sp["image_shapes"]))
    for key, lst in zip(("boxes", "scores", "labels", "all_scores", "all_boxes"), out):
        assert [int(x.shape[0]) for x in lst] == exp[key + "_n"].tolist(), key
        got = torch.cat([x.reshape(-1) for x in lst]).numpy()
        assert np.array_equal(got.reshape(-1), exp[key].reshape(-1)), key


def product_rpn(sp, device="cpu"):
    import snn_automotive_object_detection_amd as S
    from snn_automotive_object_detection_amd.stock.anchors import AnchorGenerator, ImageList
    obj, dl = FX.rpn_post_inputs(sp)
    obj, dl = [o.to(device) for o in obj], [d.to(device) for d in dl]
    N = len(sp["image_sizes"])
    rpn = S.RegionProposalNetwork(AnchorGenerator(FX.ANCHOR_SIZES, FX.ASPECT_RATIOS), StoredHead(obj, dl), 0.7, 0.3, 256, 0.5,
                                  dict(training=2000, testing=sp["pre"]), dict(training=2000, testing=sp["post"]), sp["nms"],
                                  score_thresh=sp["score_thresh"]).eval()
    images = ImageList(torch.zeros((N, 3) + tuple(sp["canvas"]), device=device), list(sp["image_sizes"]))
    feats = {str(l): torch.zeros((N, 1) + tuple(g), device=device) for l, g in enumerate(sp["grids"])}
    return rpn, images, feats


def check_rpn_against_fixture(name, boxes, pre, scores=None):
    exp = FX.load_expected(name)
    eb, es = split_rows(exp["boxes"].reshape(-1, 4), exp["boxes_n"]), split_rows(exp["scores"].reshape(-1), exp["scores_n"])
    for i in range(len(eb)):
        # the product's forward returns boxes only (like the reference's): scores are implied by the order
        sc = es[i] if scores is None else scores[i].detach().cpu().numpy()
        assert_same_detections(boxes[i].detach().cpu().numpy(), sc, eb[i], es[i], what="%s image %d" % (name, i))
        assert_same_detections(pre[i]["proposals"].detach().cpu().numpy(), pre[i]["objectness"].detach().cpu().numpy(),
                               exp["pre_boxes"][i], exp["pre_prob"][i], what="%s image %d pre-NMS" % (name, i))


@pytest.mark.parametrize("name", sorted(FX.RPN_POST_SPECS))
def test_stock_rpn_comparator_matches_reference_fixture(name):
    rpn, images, feats = product_rpn(FX.RPN_POST_SPECS[name])
    boxes, pre = rpn(images, feats)
    check_rpn_against_fixture(name, boxes, pre)


def check_det_against_fixture(name, res):
    exp = FX.load_expected(name)
    n_img = len(exp["boxes_n"])
    eb = split_rows(exp["boxes"].reshape(-1, 4), exp["boxes_n"])
    es = split_rows(exp["scores"].reshape(-1), exp["scores_n"])
    el = split_rows(exp["labels"].reshape(-1), exp["labels_n"])
    K = FX.DET_POST_SPECS[name]["K"]
    ea = split_rows(exp["all_scores"].reshape(-1, K), exp["all_scores_n"])
    eab = split_rows(exp["all_boxes"].reshape(-1, K, 4), exp["all_boxes_n"])
    boxes, scores, labels, all_scores, all_boxes = res
    for i in range(n_img):
        lab = labels[i].detach().cpu().numpy()
        n_fg = int((el[i] > 0).sum())
        assert int((lab > 0).sum()) == n_fg, "%s image %d: %d foreground rows, expected %d" % (name, i, int((lab > 0).sum()), n_fg)
        assert (lab[:n_fg] > 0).all() and (lab[n_fg:] == 0).all()          # foreground first, then the background boxes
        b, s = boxes[i].detach().cpu().numpy(), scores[i].detach().cpu().numpy()
        assert_same_detections(b[:n_fg], s[:n_fg], eb[i][:n_fg], es[i][:n_fg], lab[:n_fg], el[i][:n_fg], "%s image %d fg" % (name, i))
        assert_same_detections(b[n_fg:], s[n_fg:], eb[i][n_fg:], es[i][n_fg:], lab[n_fg:], el[i][n_fg:], "%s image %d bg" % (name, i))
        assert labels[i].dtype == torch.int64
        if ea[i].shape[0]:
            assert np.abs(all_scores[i].detach().cpu().numpy() - ea[i]).max() <= 2e-6
            assert np.abs(all_boxes[i].detach().cpu().numpy() - eab[i]).max() <= 1e-3
        else:
            assert all_scores[i].shape[0] == 0 and all_boxes[i].shape[0] == 0


@pytest.mark.parametrize("name", sorted(FX.DET_POST_SPECS))
def test_stock_det_comparator_matches_reference_fixture(name):
    import snn_automotive_object_detection_amd as S
    sp = FX.DET_POST_SPECS[name]
    logits, reg, props = FX.det_post_inputs(sp)
    heads = S.RoIHeadsSNN(None, None, 0.5, 0.5, 512, 0.25, None, 0.4, 0.5, 100)
    res = heads.postprocess_detections_reference(logits, reg, props, list(sp["image_shapes"]))
    check_det_against_fixture(name, res)
