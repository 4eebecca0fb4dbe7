"""Oracle restatement of model assembly (TEST INFRASTRUCTURE ONLY).

Follows `/root/reference/src/dsnt/model.py`: `HumanPoseModel` :21-76,
`ResNetHumanPoseModel` :79-201, `HourglassHumanPoseModel` :204-314, builders
:317-379.  The `gauss` output strategy (heat-map matching: `model.py:147-156, 247-258`) uses
`util.encode_heatmaps` / `decode_heatmaps` (restated in `dsnt_oracle/util.py`); the
reference hard-codes `.cuda()` for the targets there (`:154, :253`) — here they
follow the heat-maps' device.
"""

import inspect
import re

import torch
from torch import nn
import torch.nn.functional as F

from . import nn as onn
from . import hourglass as ohg
from . import resnet as oresnet
from . import util as outil


class ImageSpecs:
    """Input-image contract (`/root/reference/src/dsnt/data.py:19-36`), values only."""

    def __init__(self, size, subtract_mean, divide_stddev):
        self.size = size
        self.subtract_mean = subtract_mean
        self.divide_stddev = divide_stddev


def hm_preact(x, preact):
    """Normalise each [H, W] map into a distribution (model.py:24-45)."""
    n_chans, h, w = x.size(-3), x.size(-2), x.size(-1)
    flat = x.reshape(-1, h * w)
    if preact == 'softmax':
        flat = F.softmax(flat, dim=-1)
    elif preact == 'thresholded_softmax':
        flat = onn.thresholded_softmax(flat, -0.5)
    elif preact in ('abs', 'relu', 'sigmoid'):
        flat = {'abs': torch.abs, 'relu': F.relu, 'sigmoid': torch.sigmoid}[preact](flat)
        flat = flat / (flat.sum(-1, keepdim=True) + 1e-12)
    else:
        raise Exception('unrecognised heatmap preactivation function: {}'.format(preact))
    return flat.view(-1, n_chans, h, w)


def calculate_reg_loss(target, mask, reg, hm, hm_sigma):
    """Regulariser dispatch; sigma is converted pixels -> normalised (model.py:47-63)."""
    sigma = 2.0 * hm_sigma / hm.size(-1)
    fn = {'var': onn.variance_reg_loss, 'kl': onn.kl_reg_loss,
          'js': onn.js_reg_loss, 'mse': onn.mse_reg_loss}.get(reg)
    return 0 if fn is None else fn(hm, target, sigma, mask)


class HumanPoseModel(nn.Module):
    def _hm_preact(self, x, preact):
        return hm_preact(x, preact)

    def _calculate_reg_loss(self, target_var, mask_var, reg, hm_var, hm_sigma):
        return calculate_reg_loss(target_var, mask_var, reg, hm_var, hm_sigma)


class ResNetHumanPoseModel(HumanPoseModel):
    """Fully-convolutional ResNet + 1x1 heatmap conv + DSNT (model.py:79-201)."""

    def __init__(self, resnet, n_chans=16, dilate=0, truncate=0, output_strat='dsnt',
                 preact='softmax', reg='none', reg_coeff=1.0, hm_sigma=1.0):
        super().__init__()
        self.n_chans = n_chans
        self.output_strat = output_strat
        self.preact = preact
        self.reg = reg
        self.reg_coeff = reg_coeff
        self.hm_sigma = hm_sigma
        self.heatmap_size = 7 * 2 ** max(dilate, truncate)

        groups = [resnet.layer2, resnet.layer3, resnet.layer4]
        # Dilation surgery (model.py:112-121): in the last `dilate` groups the
        # stride-2 convs become stride 1 and every *other* 3x3 conv is dilated.
        for i, group in enumerate(groups[len(groups) - dilate:]):
            d = 2 ** (i + 1)
            for m in group.modules():
                if not isinstance(m, nn.Conv2d):
                    continue
                if m.stride == (2, 2):
                    m.stride = (1, 1)
                elif m.kernel_size == (3, 3):
                    m.dilation = (d, d)
                    m.padding = ((d * 2 + 1) // 2, (d * 2 + 1) // 2)
        stem = [resnet.conv1, resnet.bn1, resnet.relu, resnet.maxpool, resnet.layer1]
        self.fcn = nn.Sequential(*(stem + groups[:len(groups) - truncate]))
        if truncate > 0:
            feats = groups[-truncate][0].conv1.in_channels
        else:
            feats = resnet.fc.in_features
        self.hm_conv = nn.Conv2d(feats, n_chans, kernel_size=1, bias=False)
        if output_strat == 'fc':
            self.out_fc = nn.Linear(self.heatmap_size * self.heatmap_size, 2)

    @property
    def image_specs(self):
        return ImageSpecs(size=224, subtract_mean=False, divide_stddev=False)

    def forward_part1(self, x):
        return self.hm_conv(self.fcn(x))

    def forward_part2(self, x):
        if self.output_strat == 'dsnt':
            x = self._hm_preact(x, self.preact)
            self.heatmaps = x
            return onn.dsnt(x)
        if self.output_strat == 'fc':
            x = self._hm_preact(x, self.preact)
            self.heatmaps = x
            return self.out_fc(x.view(-1, x.size(-2) * x.size(-1))).view(-1, self.n_chans, 2)
        self.heatmaps = x
        return x

    def forward(self, *inputs):
        return self.forward_part2(self.forward_part1(inputs[0]))

    def forward_loss(self, out_var, target_var, mask_var):
        if self.output_strat in ('dsnt', 'fc'):
            loss = onn.euclidean_loss(out_var, target_var, mask_var)
            reg = self._calculate_reg_loss(target_var, mask_var, self.reg, self.heatmaps,
                                           self.hm_sigma)
            return loss + self.reg_coeff * reg
        if self.output_strat == 'gauss':     # model.py:147-156
            target_hm = outil.encode_heatmaps(target_var, out_var.size(-1), out_var.size(-2), self.hm_sigma)
            return F.mse_loss(out_var, target_hm.to(out_var.device, out_var.dtype))
        raise Exception('invalid configuration')

    def compute_coords(self, out_var):
        if self.output_strat in ('dsnt', 'fc'):
            return out_var.detach().to('cpu', torch.float32)
        if self.output_strat == 'gauss':     # model.py:160-161
            return outil.decode_heatmaps(out_var.detach().cpu())
        raise Exception('invalid configuration')


class HourglassHumanPoseModel(HumanPoseModel):
    """Stacked hourglass + per-stack DSNT heads, summed loss (model.py:204-314)."""

    def __init__(self, hg, n_chans=16, output_strat='gauss', preact='softmax', reg='none',
                 reg_coeff=1.0, hm_sigma=1.0):
        super().__init__()
        self.hg = hg
        self.n_chans = n_chans
        self.output_strat = output_strat
        self.preact = preact
        self.reg = reg
        self.reg_coeff = reg_coeff
        self.hm_sigma = hm_sigma
        self.heatmap_size = getattr(hg, 'heatmap_size', 64)
        if output_strat == 'fc':
            self.out_fc = nn.Linear(self.heatmap_size * self.heatmap_size, 2)

    @property
    def image_specs(self):
        return ImageSpecs(size=256, subtract_mean=True, divide_stddev=False)

    @property
    def heatmaps(self):
        return self.heatmaps_array[0]

    def forward_part1(self, x):
        return self.hg(x)

    def forward_part2(self, hg_outs):
        if self.output_strat == 'gauss':
            self.heatmaps_array = hg_outs
            return hg_outs
        if self.output_strat not in ('dsnt', 'fc'):
            raise Exception('invalid configuration')
        out = []
        self.heatmaps_array = []
        for x in hg_outs:  # a bare 4-D tensor is iterated along dim 0 (inference.py:47)
            x = self._hm_preact(x, self.preact)
            self.heatmaps_array.append(x)
            if self.output_strat == 'dsnt':
                out.append(onn.dsnt(x))
            else:
                flat = x.view(-1, x.size(-2) * x.size(-1))
                out.append(self.out_fc(flat).view(-1, self.n_chans, 2))
        return out

    def forward(self, *inputs):
        return self.forward_part2(self.forward_part1(inputs[0]))

    def forward_loss(self, out_vars, target_var, mask_var):
        if self.output_strat in ('dsnt', 'fc'):
            total = 0
            for i, out_var in enumerate(out_vars):
                loss = onn.euclidean_loss(out_var, target_var, mask_var)
                reg = self._calculate_reg_loss(target_var, mask_var, self.reg,
                                               self.heatmaps_array[i], self.hm_sigma)
                total = total + loss + self.reg_coeff * reg
            return total
        if self.output_strat == 'gauss':     # model.py:247-258: summed intermediate supervision, no mask
            target_hm = outil.encode_heatmaps(target_var, out_vars[0].size(-1), out_vars[0].size(-2),
                                              self.hm_sigma)
            target_hm = target_hm.to(out_vars[0].device, out_vars[0].dtype)
            return sum(F.mse_loss(hm, target_hm) for hm in out_vars)
        raise Exception('invalid configuration')

    def compute_coords(self, out_var):
        if isinstance(out_var, list):
            out_var = out_var[-1]
        if self.output_strat in ('dsnt', 'fc'):
            return out_var.detach().to('cpu', torch.float32)
        if self.output_strat == 'gauss':     # model.py:268-269
            return outil.decode_heatmaps(out_var.detach().cpu())
        raise Exception('invalid configuration')


def _build_resnet_pose_model(base, dilate=0, truncate=0, output_strat='dsnt', preact='softmax',
                             reg='none', reg_coeff=1.0, hm_sigma=1.0):
    if base not in ('resnet18', 'resnet34', 'resnet50', 'resnet101', 'resnet152'):
        raise Exception('unsupported base model type: ' + base)
    resnet = oresnet.build_resnet(base)  # reference: pretrained=True (no network here)
    return ResNetHumanPoseModel(resnet, n_chans=16, dilate=dilate, truncate=truncate,
                                output_strat=output_strat, preact=preact, reg=reg,
                                reg_coeff=reg_coeff, hm_sigma=hm_sigma)


def _build_hg_model(base, stacks=2, blocks=1, output_strat='gauss', preact='softmax',
                    reg='none', reg_coeff=1.0, hm_sigma=1.0):
    m = re.search(r'hg(\d+)', base)
    if m is not None:
        stacks = int(m.group(1))
    elif base != 'hg':
        raise Exception('unsupported base model type: ' + base)
    hg = ohg.HourglassNet(ohg.Bottleneck, num_stacks=stacks, num_blocks=blocks)
    return HourglassHumanPoseModel(hg, n_chans=16, output_strat=output_strat, preact=preact,
                                   reg=reg, reg_coeff=reg_coeff, hm_sigma=hm_sigma)


def build_mpii_pose_model(base='resnet34', **kwargs):
    """Name -> model; kwargs are filtered by the builder's signature (model.py:364-379)."""
    if base.startswith('resnet'):
        builder = _build_resnet_pose_model
    elif base.startswith('hg'):
        builder = _build_hg_model
    else:
        raise Exception('unsupported base model type: ' + base)
    accepted = [p.name for p in inspect.signature(builder).parameters.values()
                if p.default is not inspect.Parameter.empty]
    return builder(base, **{k: kwargs[k] for k in accepted if k in kwargs})
