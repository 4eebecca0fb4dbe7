"""Oracle restatement of the PCKh accumulator (TEST INFRASTRUCTURE ONLY).

Follows `/root/reference/src/dsnt/evaluator.py:9-87`.  The reference depends on
`torchnet.meter.AverageValueMeter` (not installed here); the meter is restated as
a (sum, count) pair whose `value()` returns `(mean, None)` like the original's
`(mean, std)` tuple as far as the tests look (`tests/test_evaluator.py:37`).
The per-(batch, joint) Python loop of `add` (:69-81) is kept literally — this is
the checker, not the fast path.
"""

import torch

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


class _Mean:
    def __init__(self):
        self.reset()

    def add(self, v):
        self.total += float(v)
        self.n += 1

    def reset(self):
        self.total, self.n = 0.0, 0

    def value(self):
        return (self.total / self.n if self.n else float('nan')), None


class PCKhEvaluator:
    JOINT_NAMES = JOINT_NAMES
    JOINT_GROUPS = JOINT_GROUPS

    def __init__(self, threshold=0.5):
        self.threshold = threshold
        self.meters = {name: _Mean() for name in JOINT_NAMES + list(JOINT_GROUPS)}
        self._by_joint = {j: [self.meters[n]] for j, n in enumerate(JOINT_NAMES)}
        for group, members in JOINT_GROUPS.items():
            for name in members:
                self._by_joint[JOINT_NAMES.index(name)].append(self.meters[group])

    @staticmethod
    def calculate_pckh_distance(pred, target, ref_dist):
        return torch.dist(target, pred) / ref_dist

    def add(self, pred, target, joint_mask, head_lengths):
        for b in range(pred.size(0)):
            for j in range(pred.size(1)):
                if joint_mask[b, j] == 1:
                    d = self.calculate_pckh_distance(target[b, j], pred[b, j], head_lengths[b])
                    hit = 1 if d <= self.threshold else 0
                    for meter in self._by_joint[j]:
                        meter.add(hit)

    def reset(self):
        for m in self.meters.values():
            m.reset()
