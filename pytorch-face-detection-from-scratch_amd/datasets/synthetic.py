"""Synthetic WIDER-Face-shaped annotations: per image
n ~ U{0..max_faces} boxes (the YOLO datamodule keeps images with fewer than 3 faces, datasets/WIDERFace/
datamodule.py:102), integer w,h ~ U{8..200}, x ~ U{0..size-w}, y ~ U{0..size-h}, rows [1, x, y, w, h].
Host-side data synthesis for bench.py / tools; no network, no dataset files."""
from typing import List

import torch


def synthetic_boxes(batch: int, size: int, seed: int = 1, max_faces: int = 2) -> List[torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(batch):
        n = int(torch.randint(0, max_faces + 1, (1,), generator=g))
        rows = []
        for _k in range(n):
            w = int(torch.randint(8, 201, (1,), generator=g))
            h = int(torch.randint(8, 201, (1,), generator=g))
            x = int(torch.randint(0, size - w + 1, (1,), generator=g))
            y = int(torch.randint(0, size - h + 1, (1,), generator=g))
            rows.append([1.0, float(x), float(y), float(w), float(h)])
        out.append(torch.tensor(rows, dtype=torch.float32).reshape(-1, 5))
    return out
