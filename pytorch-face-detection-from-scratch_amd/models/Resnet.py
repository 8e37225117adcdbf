"""`Resnet` with the reference's constructor, parameter names and forward signature
(models/Resnet.py:10-99): 3x3/s2 stem, pool while H > S, 3x3 p1 head."""
import torch
import torch.nn as nn

from ..convstack import StackGeometry
from .BaseModel import BaseModel
from .PoolResnet import ResidualBlock


class Resnet(BaseModel):
    def __init__(self, filters, input_shape, num_of_patches, num_of_residual_blocks=10, probability_threshold=0.5,
                 iou_threshold=0.5, pretrained=False, output_kernel_size=3):
        super().__init__(filters, input_shape, num_of_patches=num_of_patches,
                         probability_threshold=probability_threshold, iou_threshold=iou_threshold)
        self.pretrained = pretrained
        self.conv1 = nn.Conv2d(input_shape[0], filters, kernel_size=(3, 3), stride=(2, 2), padding=1)
        self.residual_blocks = nn.Sequential(
            *[ResidualBlock(filters=filters, num_of_patches=self.num_of_patches) for _ in range(num_of_residual_blocks)])
        self.out = nn.Conv2d(filters, 5, stride=(1, 1), kernel_size=(output_kernel_size, output_kernel_size), padding=1)
        self._head = (output_kernel_size, 1)

    def _geometry(self):
        return StackGeometry("resnet", self.filters, self.input_shape[0], self.input_shape[1], self.input_shape[2],
                             self.num_of_patches, len(self.residual_blocks), 3, 2, 1, *self._head, pool_mult=1)

    def forward(self, x: torch.Tensor, predict: torch.Tensor = torch.tensor(0)):
        if predict == 1:
            x = self.forward_frames(x)                      # resize / 255 (fused into the stem for uint8 frames) + conv stack
        else:
            x = self._stack_forward(x)
        if predict == 1:
            x = self.single_non_max_suppression(x[0])
        return x
