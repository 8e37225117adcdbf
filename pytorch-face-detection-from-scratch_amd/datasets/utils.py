"""`ReduceBoundingBoxes` and `nms` with the reference's names (datasets/utils.py:95-170):
threshold -> affine decode -> xyxy -> round -> greedy NMS -> xywh, in one HIP launch per batch.
"""
import torch
import torch.nn as nn

from .. import hotpath as hp
from ..hotpath import nms  # noqa: F401  (keyword-compatible stand-in for torchvision.ops.nms)


class ReduceSSDBoundingBoxes(nn.Module):
    """datasets/utils.py:8-92: SSD decode (optionally with priors) -> threshold -> round -> NMS -> xywh."""

    def __init__(self, probability_threshold: float = 0.9, iou_threshold: float = 0.5, input_shape=(3, 320, 240),
                 patch_sizes=(60, 30, 15, 7), priors=None, with_priors=False):
        super().__init__()
        self.priors = priors                                 # (P,4) table or None = calculate_priors() inside the kernel (:31-34)
        self.probability_threshold = probability_threshold
        self.iou_threshold = iou_threshold
        self.input_shape = input_shape
        _, self.width, self.height = input_shape
        self.patch_sizes = tuple(patch_sizes)
        self.with_priors = with_priors

    def forward_batch(self, x: torch.Tensor):
        """(B,P,5) -> (rows (B,P,5) [score,x,y,w,h], counts (B,)), on the GPU, no host sync."""
        return hp.ssd_reduce_bounding_boxes(x, self.probability_threshold, self.iou_threshold, self.width, self.height,
                                            self.patch_sizes, self.with_priors, self.priors)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        rows, counts = self.forward_batch(x.unsqueeze(0))
        k = int(counts[0])
        if k == 0:
            return torch.empty(0).reshape(0, 5)
        return rows[0, :k]


class ReduceBoundingBoxes(nn.Module):
    def __init__(self, probability_threshold: float = 0.9, iou_threshold: float = 0.5,
                 input_shape=(3, 320, 240), num_of_patches=40):
        super().__init__()
        self.probability_threshold = probability_threshold
        self.iou_threshold = iou_threshold
        self.input_shape = input_shape
        _, self.width, self.height = input_shape              # names as in the reference (:107)
        self.x_patch_size = self.width / num_of_patches
        self.y_patch_size = self.height / num_of_patches
        self.num_of_patches = num_of_patches

    def forward_batch(self, x: torch.Tensor):
        """(B,5,S,S) -> (rows (B,S*S,5) [score,x,y,w,h], counts (B,) int32), all on the GPU,
        no host synchronisation."""
        return hp.reduce_bounding_boxes(x, self.probability_threshold, self.iou_threshold, self.width, self.height)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """(5,S,S) -> (K',5).  No boxes -> empty (0,5) tensor on the CPU, as the reference
        returns (:170)."""
        rows, counts = self.forward_batch(x.unsqueeze(0))
        k = int(counts[0])
        if k == 0:
            return torch.empty(0).reshape(0, 5)
        return rows[0, :k]


def convert_bbx_to_xyxy(bbx):
    return bbx[0], bbx[1], bbx[0] + bbx[2], bbx[1] + bbx[3]
