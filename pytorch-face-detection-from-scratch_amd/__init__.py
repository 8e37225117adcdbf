"""MI355X-native YOLO face-detection hot path (training step + inference) behind the Python
surface of smpurkis/PyTorch-Face-Detection-from-Scratch: `models.ModelMeta`,
`models.BaseModel`, `models.PoolResnet.PoolResnet`, `models.Resnet.Resnet`,
`losses.YoloLoss.yolo_loss`, `datasets.utils.ReduceBoundingBoxes`.

All arithmetic runs in hand-written HIP kernels for gfx950 reached through the C-ABI of
include/fdet.h (ctypes, `_native.py`).  There is no CPU / eager fallback.

The directory name contains '-', so import it as `import fdet_amd` (alias module at the repo
root) or `importlib.import_module("pytorch-face-detection-from-scratch_amd")`.
"""
from . import _native  # noqa: F401
from ._native import FdetError, build  # noqa: F401

__version__ = "0.1.0"
