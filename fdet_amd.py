"""Importable alias of the package directory `pytorch-face-detection-from-scratch_amd/`
(its name is not a valid Python identifier).  `import fdet_amd` == that package; submodules
resolve to the same objects under either name (see the package's alias finder)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("pytorch-face-detection-from-scratch_amd")
sys.modules[__name__] = _pkg
