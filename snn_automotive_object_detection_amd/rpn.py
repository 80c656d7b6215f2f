"""Drop-in for the reference's ``RPNHeadSNN`` (/root/reference/rpn.py:33-121) on the MI355X kernels.

Same constructor, same ``forward`` signature and return values, same ``state_dict`` keys
(``shared_conv.weight``, ``conv_cls.weight``, ``conv_bbox.weight``) and weight init (rpn.py:78-82);
``RegionProposalNetwork.forward`` (rpn.py:613) can call it unchanged.  The spike-rate variant the
reference keeps in a string literal (rpn.py:126-200, enabled there by editing the source) is the
attribute ``spike_rates`` here.  Inference only (no autograd through the kernels)."""
from typing import List, Tuple

import torch
from torch import nn, Tensor

from . import ops


class _WeightCache:
    """packed-weight cache keyed on the parameter's storage and version counter"""
    def __init__(self):
        self.key = None
        self.val = None

    def get(self, tensors, fn):
        key = tuple((t.data_ptr(), t._version, str(t.device)) for t in tensors)
        if key != self.key:
            self.val = fn(*tensors)
            self.key = key
        return self.val


class RPNHeadSNN(nn.Module):
    """
    Spiking RPN head: per FPN level, ``num_steps`` x { LIF current encoder -> 3x3 conv -> LIF ->
    {1x1 conv -> LI (objectness), 1x1 conv -> LI (box deltas)} }; returns the last-step LI membranes.

    Args (rpn.py:45):
        in_channels (int): number of channels of the input feature
        num_anchors (int): number of anchors to be predicted
        num_steps (int): simulation time steps (T_rpn)
    """

    _version = 2

    def __init__(self, in_channels: int, num_anchors: int, num_steps) -> None:
        super().__init__()
        self.num_steps = num_steps                                    # rpn.py:52
        self.dt = 0.001                                               # rpn.py:55
        self.p_enc = ops.LIFParameters(v_th=torch.tensor(0.25))       # rpn.py:58
        self.p_lif = ops.LIFParameters(alpha=100, v_th=torch.tensor(0.1))   # rpn.py:67
        self.in_channels = in_channels
        self.num_anchors = num_anchors
        self.li_order = "jump_first"      # Norse LI update order
        self.spike_rates = False          # True: forward returns the rpn.py:126-200 third value too
        # parameters: identical modules so that state_dict keys/shapes match the reference
        self.shared_conv = nn.Conv2d(in_channels, in_channels, kernel_size=(3, 3), stride=(1, 1),
                                     padding=1, bias=False)
        self.conv_cls = nn.Conv2d(in_channels, num_anchors, kernel_size=(1, 1), stride=(1, 1), bias=False)
        self.conv_bbox = nn.Conv2d(in_channels, num_anchors * 4, kernel_size=(1, 1), stride=(1, 1), bias=False)
        for layer in self.modules():                                  # rpn.py:78-82
            if isinstance(layer, nn.Conv2d):
                torch.nn.init.normal_(layer.weight, std=0.01)
        self._cache_shared = _WeightCache()
        self._cache_heads = _WeightCache()

    def _params(self):
        return ops.make_params(self.p_enc, self.p_lif, self.dt, self.li_order)

    @torch.no_grad()
    def forward(self, x: List[Tensor]) -> Tuple[List[Tensor], List[Tensor]]:
        C, A, T = self.in_channels, self.num_anchors, int(self.num_steps)
        w_shared = self._cache_shared.get((self.shared_conv.weight,), ops.pack_conv3x3)
        w_heads = self._cache_heads.get((self.conv_cls.weight, self.conv_bbox.weight), ops.pack_heads)
        out_l, out_b, rows, (counts, sum_l, sum_b) = ops.rpn_head_forward(
            list(x), C, A, T, self._params(), w_shared, w_heads, spike_rates=self.spike_rates)
        logits, bbox_reg, rates = [], [], []
        pos = 0
        for l, f in enumerate(x):
            N, H, W = f.shape[0], f.shape[2], f.shape[3]
            n = rows[l]
            # physically NHWC; the NCHW view is what the reference returns (rpn.py:118-119) and makes
            # concat_box_prediction_layers' view/permute/reshape (rpn.py:256-258) copy-free
            logits.append(out_l[pos:pos + n].view(N, H, W, A).permute(0, 3, 1, 2))
            bbox_reg.append(out_b[pos:pos + n].view(N, H, W, 4 * A).permute(0, 3, 1, 2))
            if self.spike_rates:                                      # rpn.py:171-195
                dev = f.device
                r_sh = (counts[l, :N].to(torch.float64) / float(T * C * H * W)).to(torch.float32).view(N, 1)
                r_ob = (sum_l[pos:pos + n].view(N, -1) / T).mean(dim=1, keepdim=True)
                r_bb = (sum_b[pos:pos + n].view(N, -1) / T).mean(dim=1, keepdim=True)
                fl = lambda v: torch.tensor([v], device=dev).repeat(N, 1)
                rates += [torch.hstack((r_sh, fl(9 * (H * W) * C * C))),
                          torch.hstack((r_ob, fl(1 * (H * W) * C * A * 4))),   # labels swapped in the
                          torch.hstack((r_bb, fl(1 * (H * W) * C * A)))]       # reference; kept as is
            pos += n
        if self.spike_rates:
            return logits, bbox_reg, rates
        return logits, bbox_reg
