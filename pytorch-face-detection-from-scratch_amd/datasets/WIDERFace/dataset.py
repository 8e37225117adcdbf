"""Grid-cell target encode behind the reference's names
(WIDERFaceDataset.convert_bbx_to_feature_map, datasets/WIDERFace/dataset.py:12-64), on the GPU.

`WIDERFaceDataset` keeps the reference's constructor and the METHOD form
`convert_bbx_to_feature_map(self, bbx, img_size)`, so a caller written against the reference drops in;
`encode_batch` is the batched form the training loop uses (one launch for the whole minibatch).  The
rest of the dataset class (JPEG decode, albumentations, __getitem__) is out of scope."""
import torch

from ... import hotpath as hp


def encode_batch(boxes, img_size, num_of_patches):
    """list of (n_i,5) [conf,x,y,w,h] -> (B,5,S,S) targets in one launch."""
    return hp.encode_targets(boxes, img_size, num_of_patches)


def convert_bbx_to_feature_map(bbx: torch.Tensor, img_size, num_of_patches: int) -> torch.Tensor:
    """(n,5) -> (5,S,S).  Free-function form (self.num_of_patches is the third argument)."""
    return hp.encode_targets([bbx], img_size, num_of_patches)[0]


class WIDERFaceDataset:
    """Constructor and encode method of the reference class (dataset.py:12-64).  Only the target encode is
    implemented: indexing raises (the image pipeline is not part of the hot path)."""

    def __init__(self, data_dir, num_of_patches, input_shape, targets=None, split: str = "train", transform=None):
        self.data_dir = data_dir
        self.transform = transform
        self.targets = targets
        self.num_of_patches = num_of_patches
        self.input_shape = input_shape

    def __len__(self):
        return len(self.targets)

    def convert_bbx_to_feature_map(self, bbx, img_size):
        """bbx (n,5) rows [conf,x,y,w,h], img_size = (width, height) -> (5,S,S) on the GPU (dataset.py:32-64)."""
        return hp.encode_targets([bbx], img_size, self.num_of_patches)[0]

    def __getitem__(self, index):
        raise NotImplementedError("JPEG decode / augmentation are outside the MI355X hot path (SURVEY.md section 8): "
                                  "feed decoded uint8 frames through datasets.feed.U8BatchFeeder")
