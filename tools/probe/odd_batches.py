"""Odd batch sizes through the training step and the frame path (development probe)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import fdet_amd
from fdet_amd import hotpath as hp
from fdet_amd.datasets.synthetic import synthetic_boxes
from fdet_amd.models import ModelMeta
from fdet_amd.models.PoolResnet import PoolResnet
from fdet_amd.models.Resnet import Resnet
dev = torch.device("cuda", 0)
for name, ctor, size, S, batches in (("PoolResnet", lambda: PoolResnet(filters=64, input_shape=(3, 480, 480), num_of_patches=10), 480, 10, (1, 7, 300)),
                                     ("Resnet", lambda: Resnet(filters=64, input_shape=(3, 320, 320), num_of_patches=10, num_of_residual_blocks=6), 320, 10, (1, 5, 33))):
    model = ctor().to(dev).train()
    mm = ModelMeta(model=model, lr=1e-4); mm.configure_optimizers()
    for B in batches:
        x = torch.rand(B, 3, size, size).to(dev)
        y = hp.encode_targets(synthetic_boxes(B, size, seed=B), (size, size), S, device=dev)
        l1 = float(mm.fused_train_step(x, y)[0]); l2 = float(mm.fused_train_step(x, y)[0])
        assert l1 == l1 and l2 == l2, (name, B, l1, l2)
        print(name, "train", B, round(l1, 4), round(l2, 4))
    model.eval()
    with torch.no_grad():
        for B in batches:
            fr = torch.randint(0, 256, (B, 3, size, size), dtype=torch.uint8).to(dev)
            a = model.forward_frames(fr); b = model._stack_forward(model._preprocess(fr))
            assert torch.equal(a, b), (name, B)
            outs = model.non_max_suppression(a)
            assert len(outs) == B
            print(name, "frames", B, "ok", sum(int(o.shape[0]) for o in outs), "boxes")
