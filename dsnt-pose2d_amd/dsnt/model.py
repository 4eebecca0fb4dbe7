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
from . import hourglass
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
            raise NotImplementedError("dsnt: output_strat='fc' is not on the DSNT hot path "
                                      '(SURVEY.md §8 f-4); not implemented on the HIP path')
        raise Exception('invalid configuration')

    def forward(self, *inputs):
        return self.forward_part2(self.forward_part1(inputs[0]))

    def forward_loss(self, out_vars, target_var, mask_var):
        if self.output_strat == 'dsnt' or self.output_strat == 'fc':
            total_loss = 0
            for i, out_var in enumerate(out_vars):
                fused = self._fused.get(id(out_var))
                if fused is not None and fused[2] is out_var:
                    logits, hm, coords = fused
                    sigma = 2.0 * self.hm_sigma / hm.size(-1)
                    total_loss = total_loss + dnn.head_loss(
                        logits, hm.detach(), coords.detach(), target_var, mask_var, self.reg,
                        sigma, self.reg_coeff)
                else:
                    loss = dnn.euclidean_loss(out_var, target_var, mask_var)
                    reg_loss = self._calculate_reg_loss(
                        target_var, mask_var, self.reg, self.heatmaps_array[i], self.hm_sigma)
                    total_loss = total_loss + loss + self.reg_coeff * reg_loss
            return total_loss
        if self.output_strat == 'gauss':
            raise NotImplementedError("dsnt: output_strat='gauss' (heat-map MSE baseline) is not "
                                      'on the DSNT hot path (SURVEY.md §2 #9)')
        raise Exception('invalid configuration')

    def compute_coords(self, out_var):
        if isinstance(out_var, list):
            out_var = out_var[-1]
        if self.output_strat == 'dsnt' or self.output_strat == 'fc':
            return out_var.detach().to('cpu', torch.float32)
        if self.output_strat == 'gauss':
            raise NotImplementedError("dsnt: output_strat='gauss' decoding is not on the DSNT hot path")
        raise Exception('invalid configuration')


def _build_resnet_pose_model(base, dilate=0, truncate=0, output_strat='dsnt', preact='softmax',
                             reg='none', reg_coeff=1.0, hm_sigma=1.0):
    if base not in ('resnet18', 'resnet34', 'resnet50', 'resnet101', 'resnet152'):
        raise Exception('unsupported base model type: ' + base)
    raise NotImplementedError(
        'dsnt: the ResNet backbone has no HIP path yet (BASELINE config 1 is the CPU reference '
        'path; GPU ResNet is SURVEY.md §8 f-4). Use an hourglass base (hg1/hg2/hg8).')


def _build_hg_model(base, stacks=2, blocks=1, output_strat='gauss', preact='softmax',
                    reg='none', reg_coeff=1.0, hm_sigma=1.0):
    m = re.search(r'hg(\d+)', base)
    if m is not None:
        stacks = int(m.group(1))
    elif base == 'hg':
        pass
    else:
        raise Exception('unsupported base model type: ' + base)
    hg = hourglass.HourglassNet(hourglass.Bottleneck, num_stacks=stacks, num_blocks=blocks)
    return HourglassHumanPoseModel(hg, n_chans=16, output_strat=output_strat, preact=preact,
                                   reg=reg, reg_coeff=reg_coeff, hm_sigma=hm_sigma)


def build_mpii_pose_model(base='resnet34', **kwargs):
    """Create a pose estimation model"""
    if base.startswith('resnet'):
        build_model = _build_resnet_pose_model
    elif base.startswith('hg'):
        build_model = _build_hg_model
    else:
        raise Exception('unsupported base model type: ' + base)
    # Filter out unexpected parameters (reference model.py:374-377)
    func_params = inspect.signature(build_model).parameters.values()
    param_names = [p.name for p in func_params if p.default != inspect.Parameter.empty]
    kwargs = {k: kwargs[k] for k in param_names if k in kwargs}
    return build_model(base, **kwargs)
