"""CPU oracle for the YOLO face-detection hot path.  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement of the reference's algorithm
(smpurkis/PyTorch-Face-Detection-from-Scratch).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
The product path (``pytorch-face-detection-from-scratch_amd``) never does and fails
loudly when the HIP library is missing.

Pinning: the reference has no tests of its own.  The oracle is
pinned by golden vectors generated in the build container by *importing the reference's
own Python* (``tools/make_goldens.py``; fixtures in ``tests/golden/``).  Third-party
arithmetic that is absent from ``/root/reference`` (``torchvision.ops.nms`` /
``box_iou`` 0.11.2) is restated from its published algorithm; for those two functions
parity is UNPINNED (no reference fixture exists) and the tests say so.
"""
from .yolo_oracle import *  # noqa: F401,F403
from . import ssd_oracle  # noqa: F401,E402  (SSD detection math)
