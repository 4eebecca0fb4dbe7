"""ResNet-18/34/50/101/152 feature extractor on the MI355X launch-list engine.

The reference takes its backbone from a third-party package that is absent here —
**torchvision==0.2.0** (`/root/reference/requirements.txt:10`; call sites `src/dsnt/model.py:13,
327-336`, `tests/test_model.py:12,26,42`) — so this restates the published architecture (He et al.
2015: BasicBlock for 18/34, Bottleneck for 50/101/152) with the attribute surface
`ResNetHumanPoseModel` relies on (`conv1, bn1, relu, maxpool, layer1..layer4, fc`,
`layerN[0].conv1.in_channels`; `model.py:103-128`) and torchvision's parameter names, so a
torchvision checkpoint's `state_dict` loads.  Pretrained weights cannot be downloaded (no network):
initialisation is torchvision 0.2's (He-normal fan-out convolutions, unit BN).

As in `dsnt.hourglass` the `torch.nn` sub-modules are parameter holders; `trace()` emits the HIP
launches: conv -> [BN+ReLU folded into the next conv's operand load] -> conv -> BN + identity + ReLU
(one elementwise kernel), strided convolutions' data gradients on the phase kernel (dsnt_conv_dgrad_strided), 3x3/2 max-pool.
"""
import math

import torch.nn as nn


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

    def trace(self, t, x, P):
        skip = _trace_downsample(self.downsample, t, x, P)
        y = t.conv(x, P.conv(self.conv1), want_stats=True, name='c1')
        y = t.conv(t.norm(y, P.bn(self.bn1)), P.conv(self.conv2), want_stats=True, name='c2')
        return t.bn_add_act(y, P.bn(self.bn2), skip, relu=True, name='out')


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

    def trace(self, t, x, P):
        skip = _trace_downsample(self.downsample, t, x, P)
        y = t.conv(x, P.conv(self.conv1), want_stats=True, name='c1')
        y = t.conv(t.norm(y, P.bn(self.bn1)), P.conv(self.conv2), want_stats=True, name='c2')
        y = t.conv(t.norm(y, P.bn(self.bn2)), P.conv(self.conv3), want_stats=True, name='c3')
        return t.bn_add_act(y, P.bn(self.bn3), skip, relu=True, name='out')


def _trace_downsample(ds, t, x, P):
    """The projection shortcut: 1x1 (strided) conv + BN without ReLU."""
    if ds is None:
        return x
    d = t.conv(x, P.conv(ds[0]), want_stats=True, name='ds')
    return t.bn_act(d, P.bn(ds[1]), relu=False, name='ds_bn')


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
        raise NotImplementedError('dsnt.resnet.ResNet is a parameter holder for ResNetHumanPoseModel; the '
                                  'classification head (avgpool + fc) is not on the DSNT hot path')


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


def trace_fcn(fcn, hm_conv, t, x, P):
    """`hm_conv(fcn(x))` (reference model.py:138-141) for fcn = Sequential(conv1, bn1, relu, maxpool, layer1, ...)."""
    conv1, bn1 = fcn[0], fcn[1]
    t.mark_bucket(0)        # one gradient bucket: its all-reduce starts when the stem's weight gradient is done
    stem = P.conv(conv1)
    gi = getattr(t, 'want_input_grad', False)      # d loss / d image: the plain stem convolution with its data gradient
    x = (None if gi else t.stem_s2d(x, stem)) or t.conv(x, stem, want_stats=True, need_input_grad=gi, name='stem')
    x = t.bn_act(x, P.bn(bn1), relu=True, name='stem_act')
    x = t.maxpool3s2(x, name='pool')
    for layer in list(fcn)[4:]:
        t.flush_point()         # backward: this stage's grouped weight gradients + slab sums run beside the stage before it
        for block in layer:
            x = block.trace(t, x, P)
    return t.conv(x, P.conv(hm_conv), name='hm')
