"""Grid-cell target encode with the reference's name
(WIDERFaceDataset.convert_bbx_to_feature_map, datasets/WIDERFace/dataset.py:32-64), on the GPU.
The dataset class itself (JPEG decode, albumentations) is out of scope."""
import torch

from ... import hotpath as hp


def encode_batch(boxes, img_size, num_of_patches):
    """list of (n_i,5) [conf,x,y,w,h] -> (B,5,S,S) targets in one launch."""
    return hp.encode_targets(boxes, img_size, num_of_patches)


def convert_bbx_to_feature_map(bbx: torch.Tensor, img_size, num_of_patches: int) -> torch.Tensor:
    """(n,5) -> (5,S,S).  Free-function form of the reference method (self.num_of_patches is
    the third argument)."""
    return hp.encode_targets([bbx], img_size, num_of_patches)[0]
