"""Drop-in for the reference's ``RPNHeadSNN`` (/root/reference/rpn.py:33-121) on the MI355X kernels.

Same constructor, same ``forward`` signature and return values, same ``state_dict`` keys
(``shared_conv.weight``, ``conv_cls.weight``, ``conv_bbox.weight``) and weight init (rpn.py:78-82);
``RegionProposalNetwork.forward`` (rpn.py:613) can call it unchanged.  The spike-rate variant the
reference keeps in a string literal (rpn.py:126-200, enabled there by editing the source) is the
attribute ``spike_rates`` here.  Inference only (no autograd through the kernels)."""
import threading
import warnings
from typing import List, Tuple

import torch
from torch import nn, Tensor

from . import ops
from ._cache import StreamSafeEntry


_WARN_LOCK = threading.Lock()       # one-warning-per-reason bookkeeping of the fallback gates (module-level: a lock held by an
                                    # nn.Module would make the model un-copyable / un-picklable; StreamPipeline calls forward from several threads)


class _WeightCache:
    """packed-weight cache keyed on the parameter's storage and version counter.  In-place writes through ``param.data``
    do not bump the version counter: after such an edit call the module's ``invalidate_packed_weights()`` (``_apply`` -
    .to() / .half() / .cuda() - and ``load_state_dict`` do it themselves).  Safe across host threads / HIP streams
    (``_cache.StreamSafeEntry``: filled under a lock, consumers on another stream wait for the packing kernels)."""
    def __init__(self):
        self._entry = StreamSafeEntry()

    @property
    def val(self):
        return self._entry.val

    def invalidate(self):
        self._entry.invalidate()

    def get(self, tensors, fn):
        key = tuple((t.data_ptr(), t._version, str(t.device)) for t in tensors)
        return self._entry.get(key, lambda: fn(*tensors), tensors[0].device)


def _pack_heads_unchecked(wa, wb):
    return ops.pack_heads(wa, wb, check_split=False)          # (the modules check the split themselves: _resolve_precision)


def _warn_once_box(fn):
    """cache value = [verdict, warned?] so that the fallback warning is raised once per weight version"""
    return lambda *tensors: [fn(*tensors), False]


def _strict_if_inexact(module, prec: str, box) -> str:
    problem = box[0]
    if problem is None:
        return prec
    with _WARN_LOCK:
        first, box[1] = not box[1], True
    if first:
        warnings.warn("%s: %s - running precision \"f32_strict\" (fp32 matrix cores, fp32 VALU heads) instead of \"%s\""
                      % (type(module).__name__, problem, prec), RuntimeWarning)
    return "f32_strict"


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
        # "bf16x3": bf16 matrix cores with an exact 3-way split of the fp32 weights (default);
        # "f32": fp32 matrix cores with the conv + LIF fused over T.  Both give fp32-exact contractions.
        # "mxfp6": fp4 x fp6 block-scaled matrix path, weights as 6 planes of base-32 digits (exact within 2^5 of the
        #          block maximum, else rounded at 2^-28 of it; ~1.4x faster); needs in_channels % 128 == 0, otherwise
        #          the bf16x3 kernels run.
        self.precision = "bf16x3"
        # parameters: identical modules so that state_dict keys/shapes match the reference
        self.shared_conv = nn.Conv2d(in_channels, in_channels, kernel_size=(3, 3), stride=(1, 1),
                                     padding=1, bias=False)
        self.conv_cls = nn.Conv2d(in_channels, num_anchors, kernel_size=(1, 1), stride=(1, 1), bias=False)
        self.conv_bbox = nn.Conv2d(in_channels, num_anchors * 4, kernel_size=(1, 1), stride=(1, 1), bias=False)
        for layer in self.modules():                                  # rpn.py:78-82
            if isinstance(layer, nn.Conv2d):
                torch.nn.init.normal_(layer.weight, std=0.01)
        self._cache_shared = {"f32": _WeightCache(), "bf16x3": _WeightCache(), "mxfp6": _WeightCache()}
        self._cache_heads = _WeightCache()
        self._cache_split = _WeightCache()      # None, or why these weights cannot be carried as three bf16 planes (-> "f32_strict")
        self.last_spike_counts = None

    def invalidate_packed_weights(self) -> None:
        """drop the packed (bf16x3 / mxfp6 / f32 fragment-major) copies of the weights; they are rebuilt on the next forward"""
        for c in list(self._cache_shared.values()) + [self._cache_heads, self._cache_split]:
            c.invalidate()

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self.invalidate_packed_weights()
        return out

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        self.invalidate_packed_weights()

    def _eff_precision(self) -> str:
        if self.precision == "mxfp6" and self.in_channels % 128:
            return "bf16x3"
        return self.precision

    def _params(self, precision=None):
        return ops.make_params(self.p_enc, self.p_lif, self.dt, self.li_order, precision or self._resolve_precision())

    def _resolve_precision(self) -> str:
        """``_eff_precision()`` unless a weight the kernels of that precision would split into three bf16 planes does not split
        EXACTLY (ops.check_bf16x3_split: magnitudes with bits below 2^-133, values next to FLT_MAX, NaN / infinity): then
        "f32_strict" - fp32 matrix cores and fp32 VALU heads, nothing is split - with one RuntimeWarning per weight version.
        The verdict is cached with the packed weights (one host synchronisation per weight update)."""
        prec = self._eff_precision()
        if prec == "f32_strict" or not self.shared_conv.weight.is_cuda:       # (CPU weights: forward raises anyway - no CPU path)
            return prec
        split = (self.conv_cls.weight, self.conv_bbox.weight) + ((self.shared_conv.weight,) if prec == "bf16x3" else ())
        return _strict_if_inexact(self, prec, self._cache_split.get(split, _warn_once_box(ops.split_problem)))

    def _packed_shared(self, prec=None):
        prec = prec or self._resolve_precision()
        pack = {"f32": ops.pack_conv3x3, "f32_strict": ops.pack_conv3x3, "bf16x3": lambda w: ops.pack_conv3x3_bf16x3(w, check_split=False),
                "mxfp6": ops.pack_conv3x3_mx}[prec]
        return self._cache_shared["f32" if prec == "f32_strict" else prec].get((self.shared_conv.weight,), pack)

    @torch.no_grad()
    def forward(self, x: List[Tensor]) -> Tuple[List[Tensor], List[Tensor]]:
        C, A, T = self.in_channels, self.num_anchors, int(self.num_steps)
        prec = self._resolve_precision()
        w_shared = self._packed_shared(prec)
        w_heads = self._cache_heads.get((self.conv_cls.weight, self.conv_bbox.weight), _pack_heads_unchecked)
        out_l, out_b, rows, (counts, sum_l, sum_b, rate_rows) = ops.rpn_head_forward(
            list(x), C, A, T, self._params(prec), w_shared, w_heads, spike_rates=self.spike_rates)
        logits, bbox_reg, rates = [], [], []
        pos = 0
        for l, f in enumerate(x):
            N, H, W = f.shape[0], f.shape[2], f.shape[3]
            n = rows[l]
            # physically NHWC; the NCHW view is what the reference returns (rpn.py:118-119) and makes
            # concat_box_prediction_layers' view/permute/reshape (rpn.py:256-258) copy-free
            logits.append(out_l[pos:pos + n].view(N, H, W, A).permute(0, 3, 1, 2))
            bbox_reg.append(out_b[pos:pos + n].view(N, H, W, 4 * A).permute(0, 3, 1, 2))
            if self.spike_rates:                                      # rpn.py:171-195: three [N, 2] = (rate, FLOPs) tensors per
                rates += [rate_rows[l, j, :N] for j in range(3)]      # level, finished by snn_rpn_rates (views, no torch math)
            pos += n
        if self.spike_rates:
            self.last_spike_counts = counts                           # [levels, N] int64: shared-LIF spikes (tests, energy report)
            return logits, bbox_reg, rates
        return logits, bbox_reg


# ---------------------------------------------------------------------------------------------
# Stock-torch caller of the head (inference side of /root/reference/rpn.py:299-703).  Not part of
# the accelerated path; present so that the detector runs end-to-end without torchvision.
# ---------------------------------------------------------------------------------------------
def permute_and_flatten(layer: Tensor, N: int, A: int, C: int, H: int, W: int) -> Tensor:
    return layer.view(N, -1, C, H, W).permute(0, 3, 4, 1, 2).reshape(N, -1, C)       # rpn.py:256-258


def concat_box_prediction_layers(box_cls: List[Tensor], box_regression: List[Tensor]) -> Tuple[Tensor, Tensor]:
    """rpn.py:262-296: order (image, level, y, x, anchor)"""
    cls_flat, reg_flat = [], []
    for cls_l, reg_l in zip(box_cls, box_regression):
        N, AxC, H, W = cls_l.shape
        A = reg_l.shape[1] // 4
        C = AxC // A
        cls_flat.append(permute_and_flatten(cls_l, N, A, C, H, W))
        reg_flat.append(permute_and_flatten(reg_l, N, A, 4, H, W))
    return torch.cat(cls_flat, dim=1).flatten(0, -2), torch.cat(reg_flat, dim=1).reshape(-1, 4)


class RegionProposalNetwork(nn.Module):
    """Inference side of the reference's RegionProposalNetwork (rpn.py:330-372 ctor, 563-703 forward):
    head -> anchors -> decode -> per-level top-k -> sigmoid -> clip -> remove small -> NMS per level ->
    top post_nms_top_n.  In eval mode the second return value is the list of per-image
    {'proposals', 'objectness'} dicts (rpn.py:493-499, 692-693)."""

    def __init__(self, anchor_generator, head, fg_iou_thresh, bg_iou_thresh, batch_size_per_image,
                 positive_fraction, pre_nms_top_n, post_nms_top_n, nms_thresh, score_thresh=0.0):
        super().__init__()
        from .stock.boxes import BoxCoder
        self.anchor_generator = anchor_generator
        self.head = head
        self.box_coder = BoxCoder(weights=(1.0, 1.0, 1.0, 1.0))                       # rpn.py:347
        self._pre_nms_top_n = pre_nms_top_n
        self._post_nms_top_n = post_nms_top_n
        self.nms_thresh = nms_thresh
        self.score_thresh = score_thresh
        self.min_size = 1e-3                                                          # rpn.py:371
        # proposal selection: "hip" = snn_rpn_proposals (six launches, one host sync), "batched" = stock torch ops with
        # top-k before decode, "reference" = the reference's per-image order of operations; all three give the same
        # proposals (tests/test_gpu_e2e.py).  CPU tensors always take "reference".
        self.post = "hip"
        self._warned = set()                                                          # fallback reasons already reported

    def pre_nms_top_n(self):
        return self._pre_nms_top_n["training" if self.training else "testing"]

    def post_nms_top_n(self):
        return self._post_nms_top_n["training" if self.training else "testing"]

    def filter_proposals_reference(self, proposals, objectness, image_shapes, num_anchors_per_level):
        from .stock import boxes as box_ops
        num_images = proposals.shape[0]
        device = proposals.device
        objectness = objectness.detach().reshape(num_images, -1)
        levels = torch.cat([torch.full((n,), i, dtype=torch.int64, device=device)
                            for i, n in enumerate(num_anchors_per_level)]).reshape(1, -1).expand_as(objectness)
        idx, offset = [], 0
        for ob in objectness.split(num_anchors_per_level, 1):                         # rpn.py:434-446
            k = min(self.pre_nms_top_n(), ob.shape[1])
            idx.append(ob.topk(k, dim=1)[1] + offset)
            offset += ob.shape[1]
        top = torch.cat(idx, dim=1)
        bi = torch.arange(num_images, device=device)[:, None]
        objectness, levels, proposals = objectness[bi, top], levels[bi, top], proposals[bi, top]
        prob = torch.sigmoid(objectness)
        pre_nms = [{"proposals": p, "objectness": prob[i]} for i, p in enumerate(proposals)]   # rpn.py:493-499
        final_boxes, final_scores = [], []
        for boxes, scores, lvl, shape in zip(proposals, prob, levels, image_shapes):
            boxes = box_ops.clip_boxes_to_image(boxes, shape)
            keep = box_ops.remove_small_boxes(boxes, self.min_size)
            boxes, scores, lvl = boxes[keep], scores[keep], lvl[keep]
            keep = torch.where(scores >= self.score_thresh)[0]
            boxes, scores, lvl = boxes[keep], scores[keep], lvl[keep]
            keep = box_ops.batched_nms(boxes, scores, lvl, self.nms_thresh)[: self.post_nms_top_n()]
            final_boxes.append(boxes[keep])
            final_scores.append(scores[keep])
        return final_boxes, final_scores, pre_nms

    def filter_proposals(self, objectness, pred_bbox_deltas, anchors, image_shapes, num_anchors_per_level):
        """rpn.py:420-499 restructured for the GPU: per-level top-k on the logits FIRST, then decode / sigmoid / clip /
        size and score filters on the <= 1000 candidates per level only, all images batched, and one NMS launch for
        the whole batch (category = image x level).  Same proposals as `filter_proposals_reference` (tested); two
        host synchronisations instead of about ten per image."""
        from .stock import boxes as box_ops
        from . import ops
        num_images = objectness.numel() // sum(num_anchors_per_level)
        device = objectness.device
        n_levels = len(num_anchors_per_level)
        objectness = objectness.detach().reshape(num_images, -1)
        deltas = pred_bbox_deltas.detach().reshape(num_images, -1, 4)
        idx, lvl, offset = [], [], 0
        for i, ob in enumerate(objectness.split(num_anchors_per_level, 1)):           # rpn.py:434-446
            k = min(self.pre_nms_top_n(), ob.shape[1])
            idx.append(ob.topk(k, dim=1)[1] + offset)
            lvl.append(torch.full((k,), i, dtype=torch.int64, device=device))
            offset += ob.shape[1]
        top = torch.cat(idx, dim=1)                                                   # [N, K]
        levels = torch.cat(lvl)                                                       # [K]
        K = top.shape[1]
        logits = objectness.gather(1, top)
        d_sel = deltas.gather(1, top[..., None].expand(-1, -1, 4))
        a_sel = anchors[0][top]                                                       # anchors are per-shape, not per-image
        proposals = self.box_coder.decode_single(d_sel.reshape(-1, 4), a_sel.reshape(-1, 4)).view(num_images, K, 4)
        prob = torch.sigmoid(logits)
        pre_nms = [{"proposals": p, "objectness": prob[i]} for i, p in enumerate(proposals)]   # rpn.py:493-499
        hw = torch.tensor([[float(s[1]), float(s[0])] * 2 for s in image_shapes], dtype=proposals.dtype, device=device)
        boxes = torch.minimum(proposals.clamp(min=0), hw[:, None, :])                 # clip_boxes_to_image
        ws, hs = boxes[..., 2] - boxes[..., 0], boxes[..., 3] - boxes[..., 1]
        valid = (ws >= self.min_size) & (hs >= self.min_size) & (prob >= self.score_thresh)
        scores = torch.where(valid, prob, prob.new_full((), -1.0))                    # invalid boxes sort last
        keep_sorted, order_all = [], []
        for i in range(num_images):                  # NMS per image (pairs across images would be wasted IoUs)
            order, kept = ops.nms_keep_mask(boxes[i], scores[i], levels, self.nms_thresh)
            kept = kept & valid[i][order]
            kept = kept & (kept.cumsum(0) <= self.post_nms_top_n())
            keep_sorted.append(kept)
            order_all.append(order)
        keep_sorted = torch.stack(keep_sorted)                                         # [N, K] in score order
        order_all = torch.stack(order_all)
        counts = keep_sorted.sum(1).tolist()                                           # host sync 1
        sel = keep_sorted.reshape(-1).nonzero().squeeze(1)                             # host sync 2; row-major = by image
        img_of = sel // K
        pick = order_all.reshape(-1)[sel]
        final_boxes = list(boxes[img_of, pick].split(counts))
        final_scores = list(prob[img_of, pick].split(counts))
        return final_boxes, final_scores, pre_nms

    def _hip_proposals_refusal(self, objectness):
        """the limits of snn_rpn_proposals (csrc/snn_post.h), mirrored so that a configuration outside them takes the
        stock-torch path instead of raising: None if the HIP path can run, else the reason"""
        N, A = objectness[0].shape[0], objectness[0].shape[1]
        if len(objectness) > 8:
            return "%d feature levels (HIP path: <= 8)" % len(objectness)
        if N > 64:
            return "%d images per batch (HIP path: <= 64)" % N
        if A > 16:
            return "%d anchors per location (HIP path: <= 16)" % A
        for l, o in enumerate(objectness):                  # per-level index ranges (32-bit element indices, 28-bit slot keys)
            per_image = int(o.shape[1] * o.shape[2] * o.shape[3])
            if per_image * N > 0x7fffffff or per_image >= (1 << 28):
                return "level %d holds %d anchors per image (HIP path: < 2^28 per image, < 2^31 per batch)" % (l, per_image)
        k = sum(min(int(self.pre_nms_top_n()), int(o.shape[1] * o.shape[2] * o.shape[3])) for o in objectness)
        if k > 8192:
            return "%d pre-NMS candidates per image (HIP path: <= 8192)" % k
        return None

    def _proposals_hip(self, objectness, pred_bbox_deltas, images, feats):
        from . import ops
        N = objectness[0].shape[0]
        A = objectness[0].shape[1]
        lg, dl, hw, strides = [], [], [], []
        image_size = images.tensors.shape[-2:]
        for o, d in zip(objectness, pred_bbox_deltas):
            H, W = o.shape[-2:]
            # [N, A, H, W] views of position-major buffers (RPNHeadSNN): permute + reshape is copy-free
            lg.append(o.detach().permute(0, 2, 3, 1).reshape(N * H * W, A))
            dl.append(d.detach().permute(0, 2, 3, 1).reshape(N * H * W, 4 * A))
            hw.append((H, W))
            strides.append((image_size[0] // H, image_size[1] // W))
        boxes, scores, counts, pre_b, pre_p = ops.rpn_proposals(
            lg, dl, hw, strides, self.anchor_generator.cell_anchors, images.image_sizes, self.pre_nms_top_n(),
            self.post_nms_top_n(), self.nms_thresh, self.score_thresh, self.min_size)
        cnt = counts.tolist()                                                          # the one host synchronisation
        final = [boxes[i, :c] for i, c in enumerate(cnt)]
        pre_nms = [{"proposals": pre_b[i], "objectness": pre_p[i]} for i in range(N)]
        return final, pre_nms

    def forward(self, images, features, targets=None):
        if self.training:
            raise NotImplementedError("inference only: training the RPN is out of scope (DESIGN.md §7)")
        feats = list(features.values())
        head_out = self.head(feats)                                                   # rpn.py:613 (608-610 in spike-rate mode)
        objectness, pred_bbox_deltas = head_out[:2]
        # spike-rate mode: the head's third value takes the place of `losses` (rpn.py:698-701, "losses = spike_rates")
        rates = head_out[2] if len(head_out) == 3 else None
        if objectness[0].is_cuda and self.post == "hip":
            why = self._hip_proposals_refusal(objectness)
            if why is None:
                boxes, pre_nms = self._proposals_hip(objectness, pred_bbox_deltas, images, feats)
                return boxes, (pre_nms if rates is None else rates)
            with _WARN_LOCK:                                 # loud, once per reason
                first = why not in self._warned
                self._warned.add(why)
            if first:
                warnings.warn("RegionProposalNetwork: proposal selection falls back to the stock torch ops (%s)" % why, RuntimeWarning)
        anchors = self.anchor_generator(images, feats)
        num_images = len(anchors)
        num_anchors_per_level = [o.shape[1] * o.shape[2] * o.shape[3] for o in objectness]
        objectness, pred_bbox_deltas = concat_box_prediction_layers(objectness, pred_bbox_deltas)
        if objectness.is_cuda and self.post == "batched":
            boxes, scores, pre_nms = self.filter_proposals(objectness, pred_bbox_deltas, anchors, images.image_sizes,
                                                           num_anchors_per_level)
        else:
            proposals = self.box_coder.decode(pred_bbox_deltas.detach(), anchors).view(num_images, -1, 4)
            boxes, scores, pre_nms = self.filter_proposals_reference(proposals, objectness, images.image_sizes,
                                                                     num_anchors_per_level)
        return boxes, (pre_nms if rates is None else rates)
