"""``create_model`` with the reference's signature (/root/reference/model.py:7-189): a Faster R-CNN
(ResNet-50-FPN backbone, transform 768/1536) whose RPN head and RoI head are the spiking MI355X modules
built with the CLI step counts (model.py:61-68, 127-144, 186-187).  All hyper-parameters are the literals
of model.py:50-59, 98-106.  Pretrained weights cannot be downloaded here (no network): random init."""
import torch

from .faster_rcnn import FastRCNNPredictorSNNFull
from .generalized_rcnn import GeneralizedRCNN
from .roi_heads import RoIHeadsSNN
from .rpn import RPNHeadSNN, RegionProposalNetwork
from .stock.anchors import AnchorGenerator
from .stock.backbone import ResNet50FPN
from .stock.roi_align import MultiScaleRoIAlign
from .stock.transform import GeneralizedRCNNTransform


def _default_anchorgen():
    return AnchorGenerator(sizes=((32,), (64,), (128,), (256,), (512,)),
                           aspect_ratios=((0.5, 1.0, 2.0),) * 5)                    # faster_rcnn.py:31-34


def create_model(dataset_name, num_classes, rpn_snn=True, detector_snn=True, trainable_backbone_layers=0,
                 pretrained_rpn_and_detector=False, pretrained_fpn=False, num_steps_rpn=12, num_steps_detector=16,
                 only_one_bbox=False):
    if not (rpn_snn and detector_snn):
        raise NotImplementedError("only the spiking heads are provided (--rpn-snn --detector-snn); the ANN heads "
                                  "are the reference's baseline and out of scope")
    if pretrained_rpn_and_detector or pretrained_fpn:
        raise NotImplementedError("pretrained weights need network access; load a state_dict instead")
    if dataset_name == "cityscapes":                                               # model.py:22-24
        image_mean, image_std = [0.2869, 0.3251, 0.2839], [0.1870, 0.1902, 0.1872]
    else:                                                                          # torchvision defaults
        image_mean, image_std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    backbone = ResNet50FPN(256)
    out_channels = backbone.out_channels
    anchorgen = _default_anchorgen()
    rpn_head = RPNHeadSNN(out_channels, anchorgen.num_anchors_per_location()[0], num_steps=num_steps_rpn)
    rpn = RegionProposalNetwork(anchorgen, rpn_head, 0.7, 0.3, 256, 0.5,
                                dict(training=2000, testing=1000), dict(training=2000, testing=1000), 0.7,
                                score_thresh=0.0)                                  # model.py:50-59, 80-91
    box_roi_pool = MultiScaleRoIAlign(featmap_names=["0", "1", "2", "3"], output_size=7, sampling_ratio=2)
    resolution = box_roi_pool.output_size[0]
    head = FastRCNNPredictorSNNFull(out_channels * resolution ** 2, 1024, num_classes,
                                    num_steps=num_steps_detector, only_one_bbox=only_one_bbox)
    roi_heads = RoIHeadsSNN(box_roi_pool, head, 0.5, 0.5, 512, 0.25, None, 0.4, 0.5, 100)   # model.py:98-106
    transform = GeneralizedRCNNTransform(768, 1536, image_mean, image_std)         # faster_rcnn.py:163-164
    return GeneralizedRCNN(backbone, rpn, roi_heads, transform)
