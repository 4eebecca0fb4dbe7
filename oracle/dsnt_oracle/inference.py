"""CPU restatement of `generate_predictions` (`/root/reference/src/dsnt/inference.py:12-68`) — test
infrastructure only (see oracle/dsnt_oracle/__init__.py).  No `.cuda()`, no progress bar, no meters;
`MPII_Joint_Horizontal_Flips` (absent `torchdata.mpii`) restated as the standard MPII left/right swap."""
import torch
from torch.utils.data import DataLoader

HFLIP_INDICES = torch.LongTensor([5, 4, 3, 2, 1, 0, 6, 7, 8, 9, 15, 14, 13, 12, 11, 10])


def reverse_tensor(tensor, dim):
    # reference util.py:207-210
    indices = torch.arange(tensor.size(dim) - 1, -1, -1)
    return tensor.index_select(dim, indices)


def generate_predictions(model, dataset, use_flipped=True, batch_size=1):
    if use_flipped:
        assert batch_size == 1, 'test-time flip augmentation only work with batch_size=1'
    model.eval()
    loader = DataLoader(dataset, batch_size, num_workers=0)
    preds = torch.zeros(len(dataset), 16, 2, dtype=torch.float64)
    with torch.no_grad():
        for i, batch in enumerate(loader):
            n = batch['input'].size(0)
            if use_flipped:
                sample = batch['input']
                in_var = torch.cat([sample, reverse_tensor(sample, -1)], 0)
                hm_var = model.forward_part1(in_var)
                if isinstance(hm_var, list):
                    hm_var = hm_var[-1]
                hm1, hm2 = hm_var.split(1)
                hm2 = reverse_tensor(hm2, -1)
                hm2 = hm2.index_select(-3, HFLIP_INDICES)
                hm = (hm1 + hm2) / 2
                out_var = model.forward_part2(hm)
            else:
                out_var = model(batch['input'])
            coords = model.compute_coords(out_var)
            orig = torch.baddbmm(batch['transform_b'].double(), coords.double(), batch['transform_m'].double())
            preds[i * batch_size:i * batch_size + n] = orig
    return preds
