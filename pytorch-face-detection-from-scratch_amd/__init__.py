"""MI355X-native YOLO face-detection hot path (training step + inference) behind the Python
surface of smpurkis/PyTorch-Face-Detection-from-Scratch: `models.ModelMeta`,
`models.BaseModel`, `models.PoolResnet.PoolResnet`, `models.Resnet.Resnet`,
`losses.YoloLoss.yolo_loss`, `datasets.utils.ReduceBoundingBoxes`.

All arithmetic runs in hand-written HIP kernels for gfx950 reached through the C-ABI of
include/fdet.h (ctypes, `_native.py`).  There is no CPU / eager fallback.

The directory name contains '-', so import it as `import fdet_amd` (alias module at the repo
root) or `importlib.import_module("pytorch-face-detection-from-scratch_amd")`; both names
resolve to the SAME module objects (alias finder below), whichever is imported first.
"""
import importlib
import importlib.abc
import importlib.util
import sys

_NAMES = ("fdet_amd", "pytorch-face-detection-from-scratch_amd")
_CANON = __name__
_ALIASES = tuple(n for n in _NAMES if n != _CANON)


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """Maps `<alias>.x.y` onto the already-canonical `<canon>.x.y` module object."""

    def find_spec(self, fullname, path=None, target=None):
        for alias in _ALIASES:
            if fullname == alias or fullname.startswith(alias + "."):
                return importlib.util.spec_from_loader(fullname, self)
        return None

    def create_module(self, spec):
        for alias in _ALIASES:
            if spec.name == alias or spec.name.startswith(alias + "."):
                return importlib.import_module(_CANON + spec.name[len(alias):])
        raise ImportError(spec.name)

    def exec_module(self, module):
        return None


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())
for _a in _ALIASES:
    sys.modules.setdefault(_a, sys.modules[__name__])

from . import _native  # noqa: E402,F401
from ._native import FdetError, build  # noqa: E402,F401

__version__ = "0.1.0"
