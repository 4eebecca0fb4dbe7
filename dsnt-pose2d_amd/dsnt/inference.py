"""Test-time prediction with horizontal-flip augmentation.

Mirror of the reference's `generate_predictions` (`/root/reference/src/dsnt/inference.py:12-68`): same
arguments and result (a CPU DoubleTensor `[len(dataset), 16, 2]` of joint positions in original-image
pixels), same quirks — flip augmentation needs `batch_size == 1` (`:15-16`), only the LAST stack's
heat-maps are averaged (`:40-42`), the flipped half is mirrored back and its joints are swapped
left<->right (`:43-45`) before `forward_part2` turns the mean logits into coordinates (`:47-48`), and the
coordinates are mapped back with `baddbmm(transform_b, coords, transform_m)` in fp64 (`:54-57`).
The progress bar and the tele meters are out of scope; `time_meter`, if given, only needs `.add(seconds)`.
The backbone and the DSNT head run on the HIP path (eval-mode BN from the running statistics).
"""
import time

import torch
from torch.utils.data import DataLoader

# MPII_Joint_Horizontal_Flips of `torchdata.mpii` (absent third-party package, reference data.py:15,97):
# the standard MPII order r-ankle, r-knee, r-hip, l-hip, l-knee, l-ankle, pelvis, thorax, upper neck,
# head top, r-wrist, r-elbow, r-shoulder, l-shoulder, l-elbow, l-wrist with left and right swapped.
HFLIP_INDICES = torch.LongTensor([5, 4, 3, 2, 1, 0, 6, 7, 8, 9, 15, 14, 13, 12, 11, 10])


def reverse_tensor(tensor, dim):
    """`util.reverse_tensor` (reference util.py:207-210)."""
    return tensor.flip(dim if dim >= 0 else tensor.dim() + dim)


def generate_predictions(model, dataset, use_flipped=True, batch_size=1, time_meter=None):
    """Generate predictions with the model"""
    if use_flipped:
        assert batch_size == 1, 'test-time flip augmentation only work with batch_size=1'

    model.cuda()
    model.eval()

    loader = DataLoader(dataset, batch_size, num_workers=0)
    preds = torch.zeros(len(dataset), 16, 2, dtype=torch.float64)

    pos = 0
    with torch.no_grad():
        for batch in loader:
            n = batch['input'].size(0)
            start = time.perf_counter()
            if use_flipped:
                sample = batch['input']
                in_var = torch.cat([sample, reverse_tensor(sample, -1)], 0).cuda()
                hm_var = model.forward_part1(in_var)
                if isinstance(hm_var, (list, tuple)):
                    hm_var = hm_var[-1]             # just the last heat-map of a stacked hourglass
                hm1, hm2 = hm_var.split(1)
                hm2 = reverse_tensor(hm2, -1)
                hm2 = hm2.index_select(-3, HFLIP_INDICES.to(hm2.device))
                hm = (hm1 + hm2) / 2
                out_var = model.forward_part2(hm)
            else:
                out_var = model(batch['input'].cuda())
            coords = model.compute_coords(out_var)
            orig_preds = torch.baddbmm(batch['transform_b'].double(), coords.double(),
                                       batch['transform_m'].double())
            if time_meter is not None:
                time_meter.add(time.perf_counter() - start)
            preds[pos:pos + n] = orig_preds
            pos += n
    return preds
