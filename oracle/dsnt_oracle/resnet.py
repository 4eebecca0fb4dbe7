"""Oracle restatement of the torchvision ResNet the reference consumes.

TEST INFRASTRUCTURE ONLY.  The algorithm lives in a third-party dependency that
is absent from `/root/reference`: **torchvision==0.2.0** (`requirements.txt:10`,
`Dockerfile:71`), call sites `src/dsnt/model.py:13, 327-336` and
`tests/test_model.py:12,26,42`.  This is a restatement of the published ResNet
architecture (He et al. 2015; BasicBlock for 18/34, Bottleneck for 50/101/152)
with the attribute surface `ResNetHumanPoseModel` relies on: `conv1, bn1, relu,
maxpool, layer1..layer4, fc` and `layerN[0].conv1.in_channels`
(`model.py:103-128`).  The reference pins only shapes at this boundary
(`tests/test_model.py:11-37`); numeric parity of this backbone is by
construction (same `torch.nn` CPU ops).  Pretrained weights are not available
(no network): random init in the style of torchvision 0.2 (He-normal fan-out
convs, unit BN).
"""

import math

import torch.nn as nn
import torch.nn.functional as F


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        skip = x if self.downsample is None else self.downsample(x)
        y = F.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return F.relu(y + skip)


class BottleneckBlock(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        skip = x if self.downsample is None else self.downsample(x)
        y = F.relu(self.bn1(self.conv1(x)))
        y = F.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return F.relu(y + skip)


class ResNet(nn.Module):
    def __init__(self, block, layers, num_classes=1000):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.avgpool = nn.AvgPool2d(7, stride=1)
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                fan_out = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2.0 / fan_out))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def _make_layer(self, block, planes, blocks, stride=1):
        proj = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            proj = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride=stride, bias=False),
                nn.BatchNorm2d(planes * block.expansion))
        units = [block(self.inplanes, planes, stride, proj)]
        self.inplanes = planes * block.expansion
        units += [block(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*units)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        x = self.avgpool(x)
        return self.fc(x.flatten(1))


_CONFIGS = {
    'resnet18': (BasicBlock, [2, 2, 2, 2]),
    'resnet34': (BasicBlock, [3, 4, 6, 3]),
    'resnet50': (BottleneckBlock, [3, 4, 6, 3]),
    'resnet101': (BottleneckBlock, [3, 4, 23, 3]),
    'resnet152': (BottleneckBlock, [3, 8, 36, 3]),
}


def build_resnet(name):
    block, layers = _CONFIGS[name]
    return ResNet(block, layers)


def resnet18():
    return build_resnet('resnet18')


def resnet34():
    return build_resnet('resnet34')
