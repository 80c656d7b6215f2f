"""MI355X-native (gfx950) spiking RPN / RoI-head forward path of
aitor-martinez-seras/SNN-Automotive-Object-Detection — drop-in modules over hand-written HIP kernels."""
from .rpn import RPNHeadSNN                      # noqa: F401
from .faster_rcnn import FastRCNNPredictorSNNFull  # noqa: F401
from .rpn import RegionProposalNetwork            # noqa: F401
from .roi_heads import RoIHeadsSNN                # noqa: F401
from .generalized_rcnn import GeneralizedRCNN     # noqa: F401
from .model import create_model                   # noqa: F401
from .pipeline import StreamPipeline             # noqa: F401
