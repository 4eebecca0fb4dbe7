"""PCKh accumulation on the device (reference `src/dsnt/evaluator.py:9-87`).

Same class surface as the reference's `PCKhEvaluator` (`JOINT_NAMES`, `JOINT_GROUPS`, `meters`,
`add`, `reset`, `calculate_pckh_distance`), but `add` is one kernel over [B, J] instead of a
Python double loop with a D2H copy per step (`bin/train.py:376-377`); `add_normalized` also folds
in the back-projection to image space (`bin/train.py:243-258`, fp64 like the reference).
Meter values are accumulated as device tensors and only read when `.value()` is called.
"""
import torch

from ._lib import ptr, call


class _Meter:
    def __init__(self):
        self.reset()

    def reset(self):
        self.hits = None
        self.count = None

    def add_tensors(self, hits, count):
        self.hits = hits if self.hits is None else self.hits + hits
        self.count = count if self.count is None else self.count + count

    def value(self):
        if self.count is None or float(self.count) == 0:
            return float('nan'), None
        return float(self.hits) / float(self.count), None


class PCKhEvaluator:
    """Class for calculating and accumulating PCKh values."""

    JOINT_NAMES = [
        'rankle', 'rknee', 'rhip', 'lhip', 'lknee', 'lankle', 'pelvis', 'thorax',
        'upperneck', 'headtop', 'rwrist', 'relbow', 'rshoulder', 'lshoulder',
        'lelbow', 'lwrist',
    ]
    JOINT_GROUPS = {
        'ubody': {'rwrist', 'relbow', 'rshoulder', 'lshoulder', 'lelbow', 'lwrist'},
        'total_anewell': {'rankle', 'rknee', 'rhip', 'lhip', 'lknee', 'lankle',
                          'rwrist', 'relbow', 'lelbow', 'lwrist'},
        'total_mpii': set(JOINT_NAMES) - {'pelvis', 'thorax'},
        'all': set(JOINT_NAMES),
    }

    def __init__(self, threshold=0.5):
        self.threshold = threshold
        self.meters = {n: _Meter() for n in self.JOINT_NAMES + list(self.JOINT_GROUPS)}
        self._members = {g: [self.JOINT_NAMES.index(n) for n in sorted(names)]
                         for g, names in self.JOINT_GROUPS.items()}

    @staticmethod
    def calculate_pckh_distance(pred, target, ref_dist):
        return torch.dist(target, pred) / ref_dist

    def _accumulate(self, hits, valid):
        n_joints = hits.shape[1]
        hj, vj = hits.sum(0), valid.sum(0)
        for j in range(n_joints):
            name = self.JOINT_NAMES[j] if n_joints == len(self.JOINT_NAMES) else None
            if name is not None:
                self.meters[name].add_tensors(hj[j], vj[j])
        for g, idx in self._members.items():
            idx = [i for i in idx if i < n_joints]
            self.meters[g].add_tensors(hj[idx].sum(), vj[idx].sum())

    def add_normalized(self, norm_pred, norm_target, joint_mask, head_lengths, transform_m,
                       transform_b):
        """PCKh of predictions given in normalised coords: back-projected with
        `coords @ transform_m + transform_b` (fp64) on the device, then thresholded."""
        B, J = norm_pred.shape[0], norm_pred.shape[1]
        dev = norm_pred.device
        pred = norm_pred.detach().to(torch.float32).contiguous()
        target = norm_target.detach().to(device=dev, dtype=torch.float32).contiguous()
        m = transform_m.to(device=dev, dtype=torch.float64).contiguous()
        b = transform_b.to(device=dev, dtype=torch.float64).reshape(B, 2).contiguous()
        mask = joint_mask.to(device=dev, dtype=torch.float32).contiguous()
        head = head_lengths.to(device=dev, dtype=torch.float64).contiguous()
        hits = torch.empty(B, J, device=dev)
        valid = torch.empty(B, J, device=dev)
        call('dsnt_pckh', ptr(pred), ptr(target), ptr(m), ptr(b), ptr(mask), ptr(head),
             float(self.threshold), ptr(hits), ptr(valid), B, J)
        self._accumulate(hits, valid)

    def add(self, pred, target, joint_mask, head_lengths):
        """Calculate and accumulate PCKh values for batch (coords already in image space)."""
        B = pred.shape[0]
        dev = pred.device
        eye = torch.eye(2, dtype=torch.float64, device=dev).expand(B, 2, 2)
        zero = torch.zeros(B, 2, dtype=torch.float64, device=dev)
        # NaN targets of masked-out joints (tests/test_evaluator.py:27-31) are fine: the hit test
        # is gated by the mask inside the kernel
        self.add_normalized(pred, target, joint_mask, head_lengths, eye, zero)

    def reset(self):
        for m in self.meters.values():
            m.reset()
