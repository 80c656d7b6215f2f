"""MI355X-native (gfx950) spiking RPN / RoI-head forward path of
aitor-martinez-seras/SNN-Automotive-Object-Detection — drop-in modules over hand-written HIP kernels."""
from .rpn import RPNHeadSNN                      # noqa: F401
from .faster_rcnn import FastRCNNPredictorSNNFull  # noqa: F401
