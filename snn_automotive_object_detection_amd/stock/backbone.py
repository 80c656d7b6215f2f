"""ResNet-50 + FPN(256) + LastLevelMaxPool with FrozenBatchNorm2d (faster_rcnn.py:693-694), random init
(pretrained weights need the network).  Returns OrderedDict '0','1','2','3','pool' of [N,256,H_l,W_l]."""
from collections import OrderedDict

import torch
import torch.nn.functional as F
from torch import nn, Tensor


class FrozenBatchNorm2d(nn.Module):
    def __init__(self, n: int, eps: float = 1e-5):
        super().__init__()
        self.eps = eps
        self.register_buffer("weight", torch.ones(n))
        self.register_buffer("bias", torch.zeros(n))
        self.register_buffer("running_mean", torch.zeros(n))
        self.register_buffer("running_var", torch.ones(n))

    def forward(self, x: Tensor) -> Tensor:
        scale = self.weight * (self.running_var + self.eps).rsqrt()
        bias = self.bias - self.running_mean * scale
        return x * scale.reshape(1, -1, 1, 1) + bias.reshape(1, -1, 1, 1)


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
        out = F.relu(self.bn1(self.conv1(x)))
        out = F.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return F.relu(out + idt)


class ResNet50FPN(nn.Module):
    def __init__(self, out_channels: int = 256):
        super().__init__()
        self.out_channels = out_channels
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = FrozenBatchNorm2d(64)
        self.inplanes = 64
        self.layer1 = self._make(64, 3, 1)
        self.layer2 = self._make(128, 4, 2)
        self.layer3 = self._make(256, 6, 2)
        self.layer4 = self._make(512, 3, 2)
        chans = [256, 512, 1024, 2048]
        self.inner = nn.ModuleList([nn.Conv2d(c, out_channels, 1) for c in chans])
        self.layer = nn.ModuleList([nn.Conv2d(out_channels, out_channels, 3, padding=1) for _ in chans])
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight, a=1)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def _make(self, planes, blocks, stride):
        down = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False),
                             FrozenBatchNorm2d(planes * 4))
        layers = [Bottleneck(self.inplanes, planes, stride, down)]
        self.inplanes = planes * 4
        layers += [Bottleneck(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def forward(self, x: Tensor):
        x = F.max_pool2d(F.relu(self.bn1(self.conv1(x))), 3, stride=2, padding=1)
        c2 = self.layer1(x); c3 = self.layer2(c2); c4 = self.layer3(c3); c5 = self.layer4(c4)
        feats = [c2, c3, c4, c5]
        last = self.inner[3](c5)
        outs = [self.layer[3](last)]
        for i in (2, 1, 0):
            lat = self.inner[i](feats[i])
            last = lat + F.interpolate(last, size=lat.shape[-2:], mode="nearest")
            outs.insert(0, self.layer[i](last))
        outs.append(F.max_pool2d(outs[-1], 1, 2, 0))               # LastLevelMaxPool
        return OrderedDict(zip(["0", "1", "2", "3", "pool"], outs))
