"""Stock PyTorch-ROCm glue around the spiking heads (torchvision is not installed in this image):
box ops / NMS, anchors, image transform, MultiScaleRoIAlign, ResNet-50-FPN.  Semantics follow
torchvision 0.13 as used at the reference's call sites; none of this is on the
accelerated hot path."""
