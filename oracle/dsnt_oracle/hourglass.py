"""Oracle restatement of the Stacked-Hourglass backbone (TEST INFRASTRUCTURE ONLY).

Behaviour follows `/root/reference/src/dsnt/hourglass.py`:
`Bottleneck` :14-50, `Hourglass` :53-93, `HourglassNet` :96-177.  Module
attribute names are kept so `state_dict()` keys are interchangeable with the
reference's (`conv1.weight`, `layer1.0.bn1.weight`, `hg.0.hg.3.0.0.conv2.bias`,
`fc.0.1.running_mean`, `score_.0.weight`, ...), which is what lets the tests load
one set of weights into both.  Plain `torch.nn` CPU ops throughout.
"""

import torch.nn as nn
import torch.nn.functional as F

EXPANSION = 2


class Bottleneck(nn.Module):
    """Pre-activation residual unit: (BN-ReLU-1x1) (BN-ReLU-3x3) (BN-ReLU-1x1) + skip.

    hourglass.py:14-50.  All convolutions carry a bias; the skip is the input or
    a 1x1 projection of it.
    """

    expansion = EXPANSION

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.bn1 = nn.BatchNorm2d(inplanes)
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=True)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=True)
        self.bn3 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * EXPANSION, 1, bias=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        skip = x if self.downsample is None else self.downsample(x)
        y = self.conv1(F.relu(self.bn1(x)))
        y = self.conv2(F.relu(self.bn2(y)))
        y = self.conv3(F.relu(self.bn3(y)))
        return y + skip


def _residual_seq(n_units, planes):
    return nn.Sequential(*[Bottleneck(planes * EXPANSION, planes) for _ in range(n_units)])


class Hourglass(nn.Module):
    """Recursive encoder/decoder of given depth (hourglass.py:53-93).

    `self.hg[d]` holds the three residual groups of level d (skip, down, up);
    level 0 (the innermost) has a fourth, the bottom of the recursion.
    """

    def __init__(self, block, num_blocks, planes, depth):
        super().__init__()
        assert block is Bottleneck
        self.depth = depth
        levels = []
        for d in range(depth):
            groups = [_residual_seq(num_blocks, planes) for _ in range(3)]
            if d == 0:
                groups.append(_residual_seq(num_blocks, planes))
            levels.append(nn.ModuleList(groups))
        self.hg = nn.ModuleList(levels)

    def _level(self, n, x):
        groups = self.hg[n - 1]
        up1 = groups[0](x)
        low = groups[1](F.max_pool2d(x, 2, stride=2))
        low = self._level(n - 1, low) if n > 1 else groups[3](low)
        low = groups[2](low)
        return up1 + F.interpolate(low, scale_factor=2, mode='nearest')

    def forward(self, x):
        return self._level(self.depth, x)


class HourglassNet(nn.Module):
    """Stem + `num_stacks` x (Hourglass, residual, fc, score [, remaps]).

    hourglass.py:96-177.  Returns the list of un-normalised score maps, one per
    stack.  Note the `fc` head is conv -> BN -> ReLU (hourglass.py:146-153).
    """

    def __init__(self, block=Bottleneck, num_stacks=2, num_blocks=4, num_classes=16):
        super().__init__()
        assert block is Bottleneck
        self.inplanes = 64
        self.num_feats = 128
        self.num_stacks = num_stacks
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=True)
        self.bn1 = nn.BatchNorm2d(64)
        self.layer1 = self._make_residual(64, 1)
        self.layer2 = self._make_residual(self.inplanes, 1)
        self.layer3 = self._make_residual(self.num_feats, 1)
        ch = self.num_feats * EXPANSION
        self.hg = nn.ModuleList([Hourglass(block, num_blocks, self.num_feats, 4)
                                 for _ in range(num_stacks)])
        self.res = nn.ModuleList([self._make_residual(self.num_feats, num_blocks)
                                  for _ in range(num_stacks)])
        self.fc = nn.ModuleList([nn.Sequential(nn.Conv2d(ch, ch, 1, bias=True),
                                               nn.BatchNorm2d(ch), nn.ReLU())
                                 for _ in range(num_stacks)])
        self.score = nn.ModuleList([nn.Conv2d(ch, num_classes, 1, bias=True)
                                    for _ in range(num_stacks)])
        self.fc_ = nn.ModuleList([nn.Conv2d(ch, ch, 1, bias=True)
                                  for _ in range(num_stacks - 1)])
        self.score_ = nn.ModuleList([nn.Conv2d(num_classes, ch, 1, bias=True)
                                     for _ in range(num_stacks - 1)])

    def _make_residual(self, planes, blocks):
        """hourglass.py:128-141: first unit projects the skip if widths differ."""
        proj = None
        if self.inplanes != planes * EXPANSION:
            proj = nn.Sequential(nn.Conv2d(self.inplanes, planes * EXPANSION, 1, bias=True))
        units = [Bottleneck(self.inplanes, planes, 1, proj)]
        self.inplanes = planes * EXPANSION
        units += [Bottleneck(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*units)

    def forward(self, x):
        x = F.relu(self.bn1(self.conv1(x)))
        x = self.layer1(x)
        x = F.max_pool2d(x, 2, stride=2)
        x = self.layer3(self.layer2(x))
        outs = []
        for i in range(self.num_stacks):
            y = self.fc[i](self.res[i](self.hg[i](x)))
            score = self.score[i](y)
            outs.append(score)
            if i < self.num_stacks - 1:
                x = x + self.fc_[i](y) + self.score_[i](score)
        return outs
