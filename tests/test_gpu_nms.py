"""HIP batched NMS against the greedy definition (loop reference on the host)."""
import pytest
import torch

from snn_automotive_object_detection_amd.stock import boxes as B

pytestmark = pytest.mark.gpu


def _greedy(boxes, scores, idxs, thr):
    order = scores.argsort(descending=True, stable=True).tolist()
    iou = B.box_iou(boxes, boxes)
    keep = []
    while order:
        i = order.pop(0)
        keep.append(i)
        order = [j for j in order if not (iou[i, j] > thr and idxs[i] == idxs[j])]
    return torch.tensor(keep, dtype=torch.int64)


@pytest.mark.parametrize("n,ncat,thr", [(1, 1, 0.5), (63, 1, 0.5), (64, 2, 0.7), (65, 3, 0.3), (700, 5, 0.7), (2000, 1, 0.5)])
def test_hip_batched_nms_equals_greedy(gpu_device, n, ncat, thr):
    from snn_automotive_object_detection_amd import ops
    g = torch.Generator().manual_seed(n)
    xy = torch.rand((n, 2), generator=g) * 300
    wh = torch.rand((n, 2), generator=g) * 80 + 1
    boxes = torch.cat([xy, xy + wh], dim=1)
    scores = torch.rand((n,), generator=g)
    idxs = torch.randint(0, ncat, (n,), generator=g)
    exp = _greedy(boxes, scores, idxs, thr)
    got = ops.batched_nms(boxes.to(gpu_device), scores.to(gpu_device), idxs.to(gpu_device), thr).cpu()
    assert torch.equal(got, exp)
    top = ops.batched_nms(boxes.to(gpu_device), scores.to(gpu_device), idxs.to(gpu_device), thr, max_keep=7).cpu()
    assert torch.equal(top, exp[:7])
    # the stock dispatcher routes GPU tensors here and CPU tensors to the plain-torch form: same answer
    assert torch.equal(B.batched_nms(boxes.to(gpu_device), scores.to(gpu_device), idxs.to(gpu_device), thr).cpu(),
                       B.batched_nms(boxes, scores, idxs, thr))
