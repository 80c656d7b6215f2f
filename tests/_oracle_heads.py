"""test-side head modules backed by the oracle (CPU): let the stock glue run end-to-end without a GPU and
serve as the reference for the GPU end-to-end comparison."""
import torch
from torch import nn

from oracle import snn_oracle as OR


class OracleRPNHead(nn.Module):
    def __init__(self, src):
        super().__init__()
        self.w = [src.shared_conv.weight.detach().cpu(), src.conv_cls.weight.detach().cpu(), src.conv_bbox.weight.detach().cpu()]
        self.num_steps = src.num_steps

    def forward(self, x):
        l, b = OR.rpn_head_forward([f.cpu() for f in x], *self.w, self.num_steps)
        return list(l), list(b)


class OracleDetHead(nn.Module):
    def __init__(self, src):
        super().__init__()
        self.w = [src.fc6.weight.detach().cpu(), src.fc7.weight.detach().cpu(), src.cls_score.weight.detach().cpu(),
                  src.bbox_pred.weight.detach().cpu()]
        self.num_steps = src.num_steps

    def forward(self, x):
        return OR.det_head_forward(x.cpu(), *self.w, self.num_steps)
