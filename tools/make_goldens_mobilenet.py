#!/usr/bin/env python3
"""g13_mobilenet_weights.npz: the parameter tensors of the reference's shipped MobileNetV3-backbone archive
(saved_models/official/MobilenetV3Backbone/medium_model_15x15_480.pth), read as RAW STORAGE BYTES -- nothing is
unpickled or executed (the archive is TorchScript: torch.load(weights_only=True) refuses it, torch.jit.load would run its
code).  Build container only.   python tools/make_goldens_mobilenet.py

The 242 storages `data/0 .. data/241` are in state_dict order; names and shapes come from the architecture
(oracle/mobilenet_oracle.py), and every storage's element count is checked against them.  Stored as float32 (compressed,
~3.5 MB); the `num_batches_tracked` (int64) entries are kept.
No reference OUTPUT exists for this model (timm absent, archive not executable): parity stays unpinned; the fixture
pins the WEIGHTS the tests and the roofline run use."""
import os
import sys
import zipfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import mobilenet_oracle as MO  # noqa: E402

ARCHIVE = "/root/reference/saved_models/official/MobilenetV3Backbone/medium_model_15x15_480.pth"


def main():
    z = zipfile.ZipFile(ARCHIVE)
    names = [n for n in z.namelist() if "/data/" in n and n.rsplit("/", 1)[1].isdigit()]
    names.sort(key=lambda n: int(n.rsplit("/", 1)[1]))
    pn, sh = MO.param_names(), MO.param_shapes()
    assert len(names) == len(pn) == 242, (len(names), len(pn))
    out = {}
    for zn, name in zip(names, pn):
        raw = z.read(zn)
        if name.endswith("num_batches_tracked"):
            v = np.frombuffer(raw, dtype="<i8").copy()
            assert v.size == 1, (name, v.size)
            out[name] = v.reshape(())
        else:
            v = np.frombuffer(raw, dtype="<f4").copy()
            want = int(np.prod(sh[name])) if sh[name] else 1
            assert v.size == want, (name, v.size, want)
            out[name] = v.reshape(sh[name])
    dst = os.path.join(REPO, "tests", "golden", "g13_mobilenet_weights.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes;", sum(v.size for v in out.values()), "elements; trained",
          int(out["feature_extractor.1.num_batches_tracked"]), "batches")


if __name__ == "__main__":
    main()
