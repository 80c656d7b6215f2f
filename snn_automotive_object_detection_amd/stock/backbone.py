"""ResNet-50 + FPN(256) + LastLevelMaxPool with FrozenBatchNorm2d (faster_rcnn.py:693-694), random init
(pretrained weights need the network), parameter names as in torchvision 0.13.1 (`body.*`, `fpn.inner_blocks.N.0.*`).  Returns OrderedDict '0','1','2','3','pool' of [N,256,H_l,W_l]."""
from collections import OrderedDict

import torch
import torch.nn.functional as F
from torch import nn, Tensor

from .._cache import StreamSafeEntry


FUSED_FROZEN_BN = True      # GPU inference: scale/shift (+ residual) (+ ReLU) in one pass (ops.affine_act_nchw, bit-identical to the
                            # separate torch launches below); False = the plain torch sequence everywhere


class FrozenBatchNorm2d(nn.Module):
    def __init__(self, n: int, eps: float = 1e-5):
        super().__init__()
        self.eps = eps
        self.register_buffer("weight", torch.ones(n))
        self.register_buffer("bias", torch.zeros(n))
        self.register_buffer("running_mean", torch.zeros(n))
        self.register_buffer("running_var", torch.ones(n))
        self._folded = StreamSafeEntry()    # (scale, bias): the per-channel constants are frozen, computed once (safe across
                                            # host threads / HIP streams: pipeline.StreamPipeline)

    def _scale_bias(self):
        bufs = (self.weight, self.bias, self.running_mean, self.running_var)
        key = tuple((b.data_ptr(), b._version) for b in bufs)

        def fold():
            scale = self.weight * (self.running_var + self.eps).rsqrt()
            return scale, self.bias - self.running_mean * scale
        return self._folded.get(key, fold, self.weight.device)

    def _apply(self, fn, *a, **kw):         # .to() / .cuda(): new storages
        self._folded.invalidate()
        return super()._apply(fn, *a, **kw)

    def forward(self, x: Tensor, residual: Tensor = None, relu: bool = False, inplace: bool = False) -> Tensor:
        """x * scale + bias, then (+ residual), then (ReLU): torchvision's FrozenBatchNorm2d.forward followed by the
        Bottleneck tail, as separate operations or - same roundings - in one kernel.  ``inplace=True`` lets the fused kernel
        overwrite ``x`` (the callers inside this file pass a convolution's fresh output); by default ``x`` is left alone, as
        torchvision's module leaves it (forward hooks on the convolution, feature extractors)."""
        scale, bias = self._scale_bias()
        if (FUSED_FROZEN_BN and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous() and not x.requires_grad
                and (residual is None or (residual.is_contiguous() and residual.dtype == torch.float32 and not residual.requires_grad))):
            from .. import ops
            return ops.affine_act_nchw(x, scale, bias, residual, relu, out=x if inplace else None)
        out = x * scale.reshape(1, -1, 1, 1) + bias.reshape(1, -1, 1, 1)
        if residual is not None:
            out = out + residual
        return F.relu(out) if relu else out


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = FrozenBatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = FrozenBatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = FrozenBatchNorm2d(planes * 4)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.bn1(self.conv1(x), relu=True, inplace=True)            # (each convolution output is fresh: overwritten in place)
        out = self.bn2(self.conv2(out), relu=True, inplace=True)
        return self.bn3(self.conv3(out), residual=idt, relu=True, inplace=True)


class _ConvBlock(nn.Sequential):
    """torchvision 0.13's Conv2dNormActivation(norm_layer=None, activation_layer=None): a Sequential holding one biased
    Conv2d, so that the parameter keys read `<block>.0.weight` / `<block>.0.bias`"""
    def __init__(self, cin: int, cout: int, k: int):
        super().__init__(nn.Conv2d(cin, cout, k, padding=(k - 1) // 2))


class ResNetBody(nn.Module):
    """ResNet-50 trunk as torchvision's IntermediateLayerGetter exposes it (`backbone.body.*`): conv1, bn1, layer1..4"""
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = FrozenBatchNorm2d(64)
        self.inplanes = 64
        self.layer1 = self._make(64, 3, 1)
        self.layer2 = self._make(128, 4, 2)
        self.layer3 = self._make(256, 6, 2)
        self.layer4 = self._make(512, 3, 2)

    def _make(self, planes, blocks, stride):
        down = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False),
                             FrozenBatchNorm2d(planes * 4))
        layers = [Bottleneck(self.inplanes, planes, stride, down)]
        self.inplanes = planes * 4
        layers += [Bottleneck(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def forward(self, x: Tensor):
        x = F.max_pool2d(self.bn1(self.conv1(x), relu=True, inplace=True), 3, stride=2, padding=1)
        c2 = self.layer1(x); c3 = self.layer2(c2); c4 = self.layer3(c3); c5 = self.layer4(c4)
        return [c2, c3, c4, c5]


class FeaturePyramid(nn.Module):
    """FPN(256) + LastLevelMaxPool (`backbone.fpn.inner_blocks.N.0.*`, `backbone.fpn.layer_blocks.N.0.*`)"""
    _version = 2

    def __init__(self, chans, out_channels: int):
        super().__init__()
        self.inner_blocks = nn.ModuleList([_ConvBlock(c, out_channels, 1) for c in chans])
        self.layer_blocks = nn.ModuleList([_ConvBlock(out_channels, out_channels, 3) for _ in chans])

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        # checkpoints written before torchvision 0.13 name the convolutions `inner_blocks.N.weight` (no `.0`)
        version = local_metadata.get("version", None)
        if version is None or version < 2:
            for block in ("inner_blocks", "layer_blocks"):
                for i in range(len(self.inner_blocks)):
                    for t in ("weight", "bias"):
                        old, new = "%s%s.%d.%s" % (prefix, block, i, t), "%s%s.%d.0.%s" % (prefix, block, i, t)
                        if old in state_dict:
                            state_dict[new] = state_dict.pop(old)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)

    def forward(self, feats):
        last = self.inner_blocks[3](feats[3])
        outs = [self.layer_blocks[3](last)]
        for i in (2, 1, 0):
            lat = self.inner_blocks[i](feats[i])
            last = lat + F.interpolate(last, size=lat.shape[-2:], mode="nearest")
            outs.insert(0, self.layer_blocks[i](last))
        outs.append(F.max_pool2d(outs[-1], 1, 2, 0))               # LastLevelMaxPool
        return OrderedDict(zip(["0", "1", "2", "3", "pool"], outs))


class ResNet50FPN(nn.Module):
    """`backbone` of the detector with torchvision's BackboneWithFPN key layout (`body.*`, `fpn.*`), so that the reference's
    checkpoints ({"model": state_dict}, loaded strict=False at train.py:658,674) fill it completely"""
    def __init__(self, out_channels: int = 256):
        super().__init__()
        self.out_channels = out_channels
        self.body = ResNetBody()
        self.fpn = FeaturePyramid([256, 512, 1024, 2048], out_channels)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight, a=1)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def forward(self, x: Tensor):
        return self.fpn(self.body(x))
