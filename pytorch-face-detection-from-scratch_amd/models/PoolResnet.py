"""`PoolResnet` with the reference's constructor, parameter names and forward signature
(models/PoolResnet.py:11-105).  The nn.Conv2d modules only hold the parameters; the
arithmetic runs in the HIP conv-stack engine."""
import torch
import torch.nn as nn

from ..convstack import StackGeometry
from .BaseModel import BaseModel


class ResidualBlock(nn.Module):
    """Parameter holder for conv1/conv2 (3x3, pad 1); forward lives in the fused engine."""

    def __init__(self, filters, num_of_patches, dropout=0.25):
        super().__init__()
        self.num_of_patches = num_of_patches
        self.conv1 = nn.Conv2d(filters, filters, kernel_size=(3, 3), padding=1)
        self.conv2 = nn.Conv2d(filters, filters, kernel_size=(3, 3), padding=1)
        self.dropout = dropout


class PoolResnet(BaseModel):
    def __init__(self, filters, input_shape, num_of_patches, num_of_residual_blocks=10, probability_threshold=0.5,
                 iou_threshold=0.5, pretrained=False, input_kernel_size=10, input_stride=8, output_kernel_size=6,
                 output_padding=0):
        super().__init__(filters, input_shape, num_of_patches=num_of_patches,
                         probability_threshold=probability_threshold, iou_threshold=iou_threshold)
        self.pretrained = pretrained
        self.conv1 = nn.Conv2d(input_shape[0], filters, kernel_size=(input_kernel_size, input_kernel_size),
                               stride=(input_stride, input_stride), padding=input_kernel_size - input_stride)
        self.residual_blocks = nn.Sequential(
            *[ResidualBlock(filters=filters, num_of_patches=self.num_of_patches) for _ in range(num_of_residual_blocks)])
        self.out = nn.Conv2d(filters, 5, stride=(1, 1), kernel_size=(output_kernel_size, output_kernel_size),
                             padding=output_padding)
        self._stem = (input_kernel_size, input_stride, input_kernel_size - input_stride)
        self._head = (output_kernel_size, output_padding)

    def _geometry(self):
        return StackGeometry("poolresnet", self.filters, self.input_shape[0], self.input_shape[1], self.input_shape[2],
                             self.num_of_patches, len(self.residual_blocks), *self._stem, *self._head, pool_mult=2)

    def forward(self, x: torch.Tensor, predict: torch.Tensor = torch.tensor(0)):
        if predict == 1:
            x = self.forward_frames(x)                      # resize / 255 (fused into the stem for uint8 frames) + conv stack
        else:
            x = self._stack_forward(x)
        if predict == 1:
            x = self.single_non_max_suppression(x[0])      # image 0 only, as the reference (:103-104)
        return x
