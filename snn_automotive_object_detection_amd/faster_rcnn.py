"""Drop-in for the reference's ``FastRCNNPredictorSNNFull`` (/root/reference/faster_rcnn.py:414-516)
on the MI355X kernels: same constructor, ``forward`` signature, return values and ``state_dict``
keys (``fc6.weight``, ``fc7.weight``, ``cls_score.weight``, ``bbox_pred.weight``; default
``nn.Linear`` init).  ``RoIHeadsSNN.forward`` (roi_heads.py:1230) can call it unchanged.
``spike_rates = True`` gives the faster_rcnn.py:520-618 variant (returns ONLY the rate list)."""
import torch
import os

from torch import nn

from . import ops
from .rpn import _WeightCache, _pack_heads_unchecked, _strict_if_inexact, _warn_once_box


class FastRCNNPredictorSNNFull(nn.Module):
    """
    Spiking box head + predictor: ``num_steps`` x { encoder -> fc6 -> LIF -> fc7 -> LIF ->
    {cls_score -> LI, bbox_pred -> LI} } on the flattened RoI features.

    Args (faster_rcnn.py:427-429):
        in_channels (int): number of input features (C*7*7)
        representation_size (int): size of the intermediate representation
        num_classes (int): number of output classes (including background)
        num_steps (int): simulation time steps (T_det)
        only_one_bbox (bool): 4 box outputs instead of 4 per class
    """

    def __init__(self, in_channels, representation_size, num_classes, num_steps, only_one_bbox=False):
        super().__init__()
        self.num_steps = num_steps                                     # faster_rcnn.py:433
        self.dt = 0.001                                                # :436
        self.in_channels = in_channels
        self.representation_size = representation_size
        self.num_classes = num_classes
        self.p_enc = ops.LIFParameters(v_th=torch.tensor(0.25))        # :444
        self.p_lif = ops.LIFParameters(alpha=100, v_th=torch.tensor(0.1))   # :449,452
        self.li_order = "jump_first"
        self.spike_rates = False
        self.precision = "bf16x3"          # or "f32" (fp32 matrix cores) / "mxfp6" (fp4 x fp6 digit planes); see RPNHeadSNN
        self.fc6 = nn.Linear(in_channels, representation_size, bias=False)          # :448
        self.fc7 = nn.Linear(representation_size, representation_size, bias=False)  # :451
        self.cls_score = nn.Linear(representation_size, num_classes, bias=False)    # :455
        self.only_one_bbox = only_one_bbox                                           # :460-467
        self.bbox_pred = nn.Linear(representation_size, 4 if only_one_bbox else num_classes * 4, bias=False)
        self._c6 = {"f32": _WeightCache(), "bf16x3": _WeightCache(), "mxfp6": _WeightCache()}
        self._c7 = {"f32": _WeightCache(), "bf16x3": _WeightCache(), "mxfp6": _WeightCache()}
        self._ch = _WeightCache()
        self._c6p = _WeightCache()              # fc6 in the permuted reduction order (fc6_inner)
        self._cache_split = _WeightCache()      # None, or why these weights cannot be carried as three bf16 planes (-> "f32_strict")
        self.last_spike_counts = None

    def invalidate_packed_weights(self) -> None:
        """drop the packed copies of the weights (needed after in-place edits through ``param.data``, which do not bump the
        version counter the cache is keyed on); rebuilt on the next forward"""
        for c in list(self._c6.values()) + list(self._c7.values()) + [self._ch, self._c6p, self._cache_split]:
            c.invalidate()

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self.invalidate_packed_weights()
        return out

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        self.invalidate_packed_weights()

    def _eff_precision(self) -> str:
        if self.precision == "mxfp6" and (self.in_channels % 128 or self.representation_size % 128):
            return "bf16x3"
        return self.precision

    def _params(self, precision=None):
        return ops.make_params(self.p_enc, self.p_lif, self.dt, self.li_order, precision or self._resolve_precision())

    def _resolve_precision(self) -> str:
        """see RPNHeadSNN._resolve_precision: "f32_strict" (with a RuntimeWarning) where a weight does not split into three bf16
        planes exactly"""
        prec = self._eff_precision()
        if prec == "f32_strict" or not self.fc6.weight.is_cuda:       # (CPU weights: forward raises anyway - no CPU path)
            return prec
        split = (self.cls_score.weight, self.bbox_pred.weight) + ((self.fc6.weight, self.fc7.weight) if prec == "bf16x3" else ())
        return _strict_if_inexact(self, prec, self._cache_split.get(split, _warn_once_box(ops.split_problem)))

    def fc6_inner(self, prec=None) -> int:
        """49 when fc6's weights are packed in the permuted reduction order k' = bin * C + channel (include/snn_hip.h:
        snn_det_head_forward_k): bf16x3 on [C, 7, 7] inputs with C % 32 == 0 - what lets fc6's sparse period planes run on the
        structured-sparse matrix-core instruction.  0: the reference's order (SNN_FC6_PERM=0 forces it: A/B, tests)."""
        prec = prec or self._resolve_precision()
        # (mirrors the C side's gates: the permuted order needs the word-major fused bf16x3 layers - SNN_PLANES=rm, an A/B knob, switches
        # them off; any channel count that is a multiple of 32 is fine since round 5: k_permute_planes works in passes of 8 channel blocks)
        if prec != "bf16x3" or os.environ.get("SNN_FC6_PERM") == "0" or os.environ.get("SNN_PLANES") == "rm":
            return 0
        return 49 if (self.in_channels % 49 == 0 and (self.in_channels // 49) % 32 == 0) else 0

    def _packed(self, prec=None, inner=None):
        """packed fc6, fc7, LI heads for `prec` (default: the precision this forward resolves to); ``inner`` = 0: fc6 in the
        reference's reduction order whatever fc6_inner() says (the stage-level ops read un-permuted planes)"""
        prec = prec or self._resolve_precision()
        pack = {"f32": ops.pack_linear, "f32_strict": ops.pack_linear, "bf16x3": lambda w: ops.pack_linear_bf16x3(w, check_split=False),
                "mxfp6": ops.pack_linear_mx}[prec]
        slot = "f32" if prec == "f32_strict" else prec
        inner = self.fc6_inner(prec) if inner is None else inner
        if inner:
            w6 = self._c6p.get((self.fc6.weight,), lambda w: ops.pack_linear_bf16x3(w, check_split=False, inner=inner))
        else:
            w6 = self._c6[slot].get((self.fc6.weight,), pack)
        w7 = self._c7[slot].get((self.fc7.weight,), pack)
        wh = self._ch.get((self.cls_score.weight, self.bbox_pred.weight), _pack_heads_unchecked)
        return w6, w7, wh

    @torch.no_grad()
    def forward(self, x):
        T = int(self.num_steps)
        Hd, K = self.representation_size, self.num_classes
        K4 = self.bbox_pred.weight.shape[0]
        prec = self._resolve_precision()
        w6, w7, wh = self._packed(prec)
        x = x.flatten(start_dim=1)                                     # :473
        if x.shape[1] != self.in_channels:
            raise ValueError("expected %d input features, got %d" % (self.in_channels, x.shape[1]))
        out = ops.det_head_forward(x, Hd, K, K4, T, self._params(prec), w6, w7, wh, spike_rates=self.spike_rates, w6_inner=self.fc6_inner(prec))
        return self._finish(out, x.shape[0], x.device)

    @torch.no_grad()
    def forward_roialign(self, feats, scales, rois, roi_level):
        """Same head fed straight from the FPN maps: MultiScaleRoIAlign(7x7, sampling 2) is fused with the encoder
        (the [R,C,7,7] RoI features of roi_heads.py:1217 are never materialised).  rois [R,5] = (image, x1,y1,x2,y2)."""
        T = int(self.num_steps)
        Hd, K = self.representation_size, self.num_classes
        K4 = self.bbox_pred.weight.shape[0]
        prec = self._resolve_precision()
        w6, w7, wh = self._packed(prec)
        if feats[0].shape[1] * 49 != self.in_channels:
            raise ValueError("expected %d input features, got %d x 49" % (self.in_channels, feats[0].shape[1]))
        out = ops.det_head_forward_roialign(feats, scales, rois[:, 1:5], rois[:, 0], roi_level, Hd, K, K4, T,
                                            self._params(prec), w6, w7, wh, spike_rates=self.spike_rates, w6_inner=self.fc6_inner(prec))
        return self._finish(out, rois.shape[0], rois.device)

    def _finish(self, out, R, dev):
        cls, bbox, extras = out
        if not self.spike_rates:
            return cls, bbox                                           # :513-516
        # faster_rcnn.py:568-618: four [R, 2] = (rate, "FLOPs") tensors - lif6, lif7, cls_score, bbox_pred - finished by
        # snn_det_rates from the integer spike counts of the LIF epilogues and the time-summed LI membranes (one launch)
        self.last_spike_counts = (extras[0], extras[1])
        rates = ops.det_rates(extras, self.in_channels, self.representation_size, self.num_classes,
                              self.bbox_pred.weight.shape[0], int(self.num_steps), self.only_one_bbox)
        return [rates[0], rates[1], rates[2], rates[3]]
