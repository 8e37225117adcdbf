"""Counterpart of the reference's train_model.py (27-61): build PoolResnet -> ModelMeta -> fit -> to_torchscript.

The WIDER-Face datamodule of the reference downloads and augments the dataset (out of scope); without
`--data` this script trains on synthetic WIDER-Face-shaped batches (uint8 frames + encoded targets), which
is what there is on a box without network.  `--data DIR` expects `DIR/images_u8.pt` (N,3,H,W uint8) and `DIR/boxes.pt`
(list of (n_i,5) [1,x,y,w,h]) prepared offline.

    python -m fdet_amd.train_model --filters 64 --epochs 2 --batch-size 8 --steps-per-epoch 20 --save model.pt
"""
import argparse
from pathlib import Path

import torch


def synthetic_loader(n_batches, batch_size, size, S, seed):
    """Re-iterable list of (uint8 frames, encoded targets, boxes) host batches."""
    from .datasets.synthetic import synthetic_boxes
    from .datasets.WIDERFace.dataset import encode_batch
    out = []
    g = torch.Generator().manual_seed(seed)
    for b in range(n_batches):
        x = torch.randint(0, 256, (batch_size, 3, size, size), dtype=torch.uint8, generator=g)
        boxes = synthetic_boxes(batch_size, size, seed=seed * 1000 + b)
        y = encode_batch(boxes, (size, size), S).cpu()          # targets as a DataLoader hands them over: host tensors
        out.append((x, y, boxes))
    return out


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--filters", type=int, default=128)          # train_model.py:17
    ap.add_argument("--patches", type=int, default=10)
    ap.add_argument("--size", type=int, default=480)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--epochs", type=int, default=70)
    ap.add_argument("--batch-size", type=int, default=8)
    ap.add_argument("--steps-per-epoch", type=int, default=50)
    ap.add_argument("--val-steps", type=int, default=5)
    ap.add_argument("--save", default=None)
    args = ap.parse_args(argv)
    torch.random.manual_seed(0)                                  # train_model.py:13
    from .models import ModelMeta
    from .models.PoolResnet import PoolResnet
    from .trainer import fit
    name = f"custom_poolresnet_{args.filters}_{args.patches}x{args.patches}_{args.size}x{args.size}"
    log_path = Path(f"logs/out_{name}.log")
    log_path.parent.mkdir(parents=True, exist_ok=True)
    log_path.unlink(missing_ok=True)
    model = PoolResnet(filters=args.filters, input_shape=(3, args.size, args.size), num_of_patches=args.patches,
                       num_of_residual_blocks=10).cuda()
    model.summary()
    model_setup = ModelMeta(model=model, lr=args.lr, log_path=log_path)
    train = synthetic_loader(args.steps_per_epoch, args.batch_size, args.size, args.patches, seed=1)
    val = synthetic_loader(args.val_steps, args.batch_size, args.size, args.patches, seed=2)
    hist = fit(model_setup, train, val, epochs=args.epochs, torchscript_path=args.save)
    print(f"\nfinal training loss {float(hist['train'][-1]['loss']):.3f}")
    return hist


if __name__ == "__main__":
    main()
