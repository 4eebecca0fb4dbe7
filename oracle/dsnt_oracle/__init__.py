"""CPU oracle for the dsnt-pose2d hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain-PyTorch (CPU) restatement of
the reference's algorithm for the hot path (`/root/reference/src/dsnt/nn.py`,
`model.py`, `hourglass.py`, `evaluator.py`, plus the third-party torchvision
ResNet the reference consumes).  It exists so that the HIP product path under
`dsnt-pose2d_amd/` can be checked against something that is itself pinned to the
reference:

* pinned against every known-answer vector the reference's own tests hold for
  this path (`tests/test_nn.py`, `tests/test_evaluator.py`) — see
  `tests/test_oracle_known_answers.py`;
* pinned against the reference itself, imported in the build container from
  `/root/reference/src` (forward, loss and every parameter gradient) — see
  `tests/test_oracle_vs_reference.py` (skipped where the reference is absent);
* pinned against committed golden vectors generated from that import —
  `tests/golden/*.npz`, generator `tests/golden/make_golden.py`.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this package, and only as the checker / reported CPU baseline.  Nothing
under `dsnt-pose2d_amd/` imports it; the product path has no CPU fallback.
"""

from . import nn, hourglass, resnet, model, evaluator  # noqa: F401
