"""Pose-model assembly on MI355X — same surface as the reference's `dsnt.model`.

`build_mpii_pose_model(base, **kwargs)` and the model methods `forward`, `forward_part1`,
`forward_part2`, `forward_loss`, `compute_coords`, `image_specs`, `heatmaps`,
`heatmaps_array` keep the reference's names, argument meaning, quirks and error messages
(`/root/reference/src/dsnt/model.py:21-76, 204-379`), so `train.py` / `infer.py` style callers
work unchanged.  The backbone runs on the traced HIP launch lists (`dsnt.hourglass`), the head
and loss on the fused DSNT kernels (`dsnt.nn`).
"""
import inspect
import re

import torch
from torch import nn

from . import nn as dnn
from . import util as dutil
from . import hourglass
from . import resnet
from .data import ImageSpecs


class HumanPoseModel(nn.Module):
    """Abstract base class for human pose estimation models."""

    def _hm_preact(self, x, preact):
        return dnn.hm_preact(x, preact)

    def _calculate_reg_loss(self, target_var, mask_var, reg, hm_var, hm_sigma):
        # sigma: pixels -> normalised units (reference model.py:49)
        sigma = 2.0 * hm_sigma / hm_var.size(-1)
        if reg == 'var':
            return dnn.variance_reg_loss(hm_var, target_var, sigma, mask_var)
        if reg == 'kl':
            return dnn.kl_reg_loss(hm_var, target_var, sigma, mask_var)
        if reg == 'js':
            return dnn.js_reg_loss(hm_var, target_var, sigma, mask_var)
        if reg == 'mse':
            return dnn.mse_reg_loss(hm_var, target_var, sigma, mask_var)
        return 0

    @property
    def image_specs(self):
        raise NotImplementedError()

    def forward_loss(self, out_var, target_var, mask_var):
        raise NotImplementedError()

    def compute_coords(self, out_var):
        raise NotImplementedError()


class HourglassHumanPoseModel(HumanPoseModel):
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
        try:
            self.heatmap_size = hg.heatmap_size
        except AttributeError:
            self.heatmap_size = 64
        if self.output_strat == 'fc':
            self.out_fc = nn.Linear(self.heatmap_size * self.heatmap_size, 2)
        self._fused = {}     # id(coords tensor) -> (logits, heatmaps, coords) of the fused head

    @property
    def image_specs(self):
        return ImageSpecs(size=256, subtract_mean=True, divide_stddev=False)

    @property
    def heatmaps(self):
        return self.heatmaps_array[0]

    def forward_part1(self, x):
        """Forward from images to unnormalized heatmaps"""
        return self.hg(x)

    def forward_part2(self, hg_outs):
        """Forward from unnormalized heatmaps to output (list of tensors, or a bare tensor that
        is iterated along dim 0 like the reference does for test-time flip averaging)."""
        out = []
        self._fused = {}
        if self.output_strat == 'gauss':
            self.heatmaps_array = hg_outs
            return hg_outs
        self._fused_all = None
        stacked = getattr(hg_outs, 'stacked', None)
        if self.output_strat == 'dsnt' and self.preact == 'softmax' and stacked is not None and stacked.dim() == 5:
            # every stack's head in ONE launch on the slab the hourglass hands out ([S, N, J, H, W]; rows = S N J)
            S = stacked.size(0)
            x4 = stacked.reshape(-1, stacked.size(-3), stacked.size(-2), stacked.size(-1))
            hm, coords = dnn.head_forward(x4)
            out = list(coords.view(S, -1, coords.size(-2), 2).unbind(0))
            self.heatmaps_array = list(hm.view(stacked.shape).unbind(0))
            self._fused_all = (x4, hm, coords, out)
            return out
        if self.output_strat == 'dsnt':
            self.heatmaps_array = []
            for x in hg_outs:
                if self.preact == 'softmax':
                    x4 = x.reshape(-1, x.size(-3), x.size(-2), x.size(-1))
                    hm, coords = dnn.head_forward(x4)          # one fused pass
                    self._fused[id(coords)] = (x4, hm, coords)
                else:
                    hm = self._hm_preact(x, self.preact)
                    coords = dnn.dsnt(hm)
                self.heatmaps_array.append(hm)
                out.append(coords)
            return out
        if self.output_strat == 'fc':
            self.heatmaps_array = []
            for x in hg_outs:
                hm = self._hm_preact(x, self.preact)
                self.heatmaps_array.append(hm)
                out.append(dnn.fc_coords(hm, self.out_fc.weight, self.out_fc.bias).view(-1, self.n_chans, 2))
            return out
        raise Exception('invalid configuration')

    def forward(self, *inputs):
        return self.forward_part2(self.forward_part1(inputs[0]))

    def forward_loss(self, out_vars, target_var, mask_var):
        if self.output_strat == 'dsnt' or self.output_strat == 'fc':
            total_loss = 0
            denom2 = None            # the masked-average denominator: one tiny launch shared by all stacks
            fa = getattr(self, '_fused_all', None)
            if fa is not None and len(out_vars) == len(fa[3]) and all(a is b for a, b in zip(out_vars, fa[3])):
                # the stacks' losses in ONE launch: sum_s masked_average_s = (sum over all rows) / (one stack's denominator)
                x4, hm, coords, out = fa
                S, rows = len(out), coords.numel() // 2 // len(out)
                sigma = 2.0 * self.hm_sigma / hm.size(-1)
                m_ = None if mask_var is None else mask_var.to(torch.float32).expand(out[0].shape[:-1])
                denom2 = dnn.mask_denominator(m_, rows, hm.device)
                t_all = target_var.to(torch.float32).expand_as(out[0]).unsqueeze(0).expand(S, *out[0].shape).reshape(coords.shape)
                m_all = None if m_ is None else m_.unsqueeze(0).expand(S, *m_.shape).reshape(coords.shape[:-1])
                return dnn.head_loss(x4, hm.detach(), coords.detach(), t_all, m_all, self.reg, sigma, self.reg_coeff, denom2)
            if fa is not None:
                # the caller took the outputs apart: per-stack losses on views of the fused head's tensors
                x4, hm, coords, out = fa
                S = len(out)
                x5, hm5 = x4.view(S, -1, *x4.shape[1:]), hm.view(S, -1, *hm.shape[1:])
                for s_, o in enumerate(out):
                    self._fused[id(o)] = (x5[s_], hm5[s_], o)
            for i, out_var in enumerate(out_vars):
                fused = self._fused.get(id(out_var))
                if fused is not None and fused[2] is out_var:
                    logits, hm, coords = fused
                    sigma = 2.0 * self.hm_sigma / hm.size(-1)
                    if denom2 is None:
                        m_ = None if mask_var is None else mask_var.to(torch.float32).expand(coords.shape[:-1])
                        denom2 = dnn.mask_denominator(m_, coords.numel() // 2, hm.device)
                    total_loss = total_loss + dnn.head_loss(
                        logits, hm.detach(), coords.detach(), target_var, mask_var, self.reg,
                        sigma, self.reg_coeff, denom2)
                else:
                    loss = dnn.euclidean_loss(out_var, target_var, mask_var)
                    reg_loss = self._calculate_reg_loss(
                        target_var, mask_var, self.reg, self.heatmaps_array[i], self.hm_sigma)
                    total_loss = total_loss + loss + self.reg_coeff * reg_loss
            return total_loss
        if self.output_strat == 'gauss':       # model.py:247-258: summed intermediate supervision, no mask
            return sum(dutil.heatmap_mse_loss(hm, target_var, self.hm_sigma) for hm in out_vars)
        raise Exception('invalid configuration')

    def compute_coords(self, out_var):
        if isinstance(out_var, list):
            out_var = out_var[-1]
        if self.output_strat == 'dsnt' or self.output_strat == 'fc':
            return out_var.detach().to('cpu', torch.float32)
        if self.output_strat == 'gauss':       # model.py:268-269
            return dutil.decode_heatmaps(out_var).to('cpu', torch.float32)
        raise Exception('invalid configuration')


class ResNetHumanPoseModel(HumanPoseModel, hourglass.TapeModule):
    tape_exclude = ('out_fc.',)      # the head's Linear is not part of the traced backbone
    """Fully-convolutional ResNet + 1x1 heat-map conv + DSNT (reference model.py:79-201).

    Same constructor, attributes, quirks and `state_dict()` keys (`fcn.0.weight`, `fcn.4.0.conv1.weight`,
    `hm_conv.weight`, ...).  `forward_part1` runs the traced HIP launch lists of the whole FCN; outputs
    are single tensors, not lists (`:143-171`)."""
    supports_input_grad = True

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
        fcn_modules = [resnet.conv1, resnet.bn1, resnet.relu, resnet.maxpool, resnet.layer1]
        layers = [resnet.layer2, resnet.layer3, resnet.layer4]
        # dilation surgery (model.py:112-121): in the last `dilate` groups stride-2 convs become stride 1 and
        # every OTHER 3x3 conv is dilated (`elif`: the conv that lost its stride keeps dilation 1)
        for i, layer in enumerate(layers[len(layers) - dilate:]):
            d = 2 ** (i + 1)
            for module in layer.modules():
                if isinstance(module, nn.Conv2d):
                    if module.stride == (2, 2):
                        module.stride = (1, 1)
                    elif module.kernel_size == (3, 3):
                        module.dilation = (d, d)
                        module.padding = ((d * 2 + 1) // 2, (d * 2 + 1) // 2)
        fcn_modules.extend(layers[:len(layers) - truncate])
        self.fcn = nn.Sequential(*fcn_modules)
        if truncate > 0:
            feats = layers[-truncate][0].conv1.in_channels
        else:
            feats = resnet.fc.in_features
        self.hm_conv = nn.Conv2d(feats, self.n_chans, kernel_size=1, bias=False)
        self.out_channels = n_chans
        if self.output_strat == 'fc':
            self.out_fc = nn.Linear(self.heatmap_size * self.heatmap_size, 2)

    @property
    def image_specs(self):
        return ImageSpecs(size=224, subtract_mean=False, divide_stddev=False)

    def trace(self, t, x, P):
        return resnet.trace_fcn(self.fcn, self.hm_conv, t, x, P)

    def forward_part1(self, x):
        """Forward from images to unnormalized heatmaps"""
        return self._runner()(x)

    def forward_part2(self, x):
        """Forward from unnormalized heatmaps to output"""
        self._fused = {}
        if self.output_strat == 'dsnt':
            if self.preact == 'softmax':
                x4 = x.reshape(-1, x.size(-3), x.size(-2), x.size(-1))
                hm, coords = dnn.head_forward(x4)
                self._fused[id(coords)] = (x4, hm, coords)
            else:
                hm = self._hm_preact(x, self.preact)
                coords = dnn.dsnt(hm)
            self.heatmaps = hm
            return coords
        if self.output_strat == 'fc':
            hm = self._hm_preact(x, self.preact)
            self.heatmaps = hm
            return dnn.fc_coords(hm, self.out_fc.weight, self.out_fc.bias).view(-1, self.n_chans, 2)
        if self.output_strat == 'gauss':
            self.heatmaps = x
            return x
        raise Exception('invalid configuration')

    def forward(self, *inputs):
        return self.forward_part2(self.forward_part1(inputs[0]))

    def forward_loss(self, out_var, target_var, mask_var):
        if self.output_strat == 'dsnt' or self.output_strat == 'fc':
            fused = getattr(self, '_fused', {}).get(id(out_var))
            if fused is not None and fused[2] is out_var:
                logits, hm, coords = fused
                sigma = 2.0 * self.hm_sigma / hm.size(-1)
                return dnn.head_loss(logits, hm.detach(), coords.detach(), target_var, mask_var, self.reg,
                                     sigma, self.reg_coeff)
            loss = dnn.euclidean_loss(out_var, target_var, mask_var)
            reg_loss = self._calculate_reg_loss(target_var, mask_var, self.reg, self.heatmaps, self.hm_sigma)
            return loss + self.reg_coeff * reg_loss
        if self.output_strat == 'gauss':       # model.py:147-156
            return dutil.heatmap_mse_loss(out_var, target_var, self.hm_sigma)
        raise Exception('invalid configuration')

    def compute_coords(self, out_var):
        if self.output_strat == 'dsnt' or self.output_strat == 'fc':
            return out_var.detach().to('cpu', torch.float32)
        if self.output_strat == 'gauss':       # model.py:160-161
            return dutil.decode_heatmaps(out_var).to('cpu', torch.float32)
        raise Exception('invalid configuration')


def _build_resnet_pose_model(base, dilate=0, truncate=0, output_strat='dsnt', preact='softmax',
                             reg='none', reg_coeff=1.0, hm_sigma=1.0):
    """ResNet-based pose model (reference model.py:317-343).  The reference downloads torchvision's
    pretrained weights; there is no network here, so the backbone starts from torchvision's random
    initialisation — load a checkpoint with `load_state_dict` for real use."""
    if base not in ('resnet18', 'resnet34', 'resnet50', 'resnet101', 'resnet152'):
        raise Exception('unsupported base model type: ' + base)
    net = resnet.build_resnet(base)
    return ResNetHumanPoseModel(net, n_chans=16, dilate=dilate, truncate=truncate, output_strat=output_strat,
                                preact=preact, reg=reg, reg_coeff=reg_coeff, hm_sigma=hm_sigma)


def _build_hg_model(base, stacks=2, blocks=1, output_strat='gauss', preact='softmax',
                    reg='none', reg_coeff=1.0, hm_sigma=1.0):
    """Stacked-hourglass pose model (reference model.py:346-361): 'hg' keeps `stacks`, 'hg<N>' means N stacks;
    the default output strategy of THIS builder is the heat-map one ('gauss'), as in the reference."""
    suffix = re.search(r'hg(\d+)', base)
    if suffix is None and base != 'hg':
        raise Exception('unsupported base model type: ' + base)
    n_stacks = int(suffix.group(1)) if suffix is not None else stacks
    backbone = hourglass.HourglassNet(hourglass.Bottleneck, num_stacks=n_stacks, num_blocks=blocks)
    head = dict(output_strat=output_strat, preact=preact, reg=reg, reg_coeff=reg_coeff, hm_sigma=hm_sigma)
    return HourglassHumanPoseModel(backbone, n_chans=16, **head)


_BUILDERS = (('resnet', _build_resnet_pose_model), ('hg', _build_hg_model))


def build_mpii_pose_model(base='resnet34', **kwargs):
    """Create a pose estimation model (reference model.py:364-379).  Keyword arguments the chosen family's builder
    does not declare (e.g. `dilate` for an hourglass) are dropped silently, so one `model_desc` dict serves both."""
    builder = next((fn for prefix, fn in _BUILDERS if base.startswith(prefix)), None)
    if builder is None:
        raise Exception('unsupported base model type: ' + base)
    accepted = {name for name, prm in inspect.signature(builder).parameters.items()
                if prm.default is not inspect.Parameter.empty}
    return builder(base, **{k: v for k, v in kwargs.items() if k in accepted})
