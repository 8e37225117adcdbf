#!/usr/bin/env python3
"""BASELINE.json config 3 on ONE GPU: Resnet backbone (filters 64, 10 blocks) at 3x640x640, S=20,
32 images per GPU (global batch 256 on 8 GPUs); a few fused training steps, reported as imgs/s.
   python tools/run_config3.py [--batch 32] [--steps 5]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fdet_amd
from fdet_amd import hotpath as hp
from fdet_amd.models import ModelMeta
from fdet_amd.models.Resnet import Resnet
from fdet_amd.convstack import KernelTimer
import oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--size", type=int, default=640)
ap.add_argument("--patches", type=int, default=20)
args = ap.parse_args()
torch.manual_seed(0)
dev = torch.device("cuda", 0)
model = Resnet(filters=64, input_shape=(3, args.size, args.size), num_of_patches=args.patches, num_of_residual_blocks=10).to(dev).train()
mm = ModelMeta(model=model, lr=1e-4)
mm.configure_optimizers()
g = torch.Generator().manual_seed(3)
x = torch.rand(args.batch, 3, args.size, args.size, generator=g).to(dev)
y = hp.encode_targets(O.synthetic_boxes(args.batch, args.size, seed=4), (args.size, args.size), args.patches, device=dev)
for _ in range(2):
    lsum, _, _ = mm.fused_train_step(x, y)
torch.cuda.synchronize()
timer = KernelTimer(); model.engine.timer = timer
t0 = time.perf_counter()
for _ in range(args.steps):
    lsum, yh, _ = mm.fused_train_step(x, y)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.steps
model.engine.timer = None
per = timer.summary()
print(json.dumps({"config": f"Resnet-64 {args.size}^2 S={args.patches} bs={args.batch}, 1 GPU, fwd+loss+bwd+Adam", "ms_per_step": round(dt * 1e3, 2),
                  "imgs_per_s": round(args.batch / dt, 1), "loss": float(lsum), "finite": bool(torch.isfinite(yh).all()),
                  "kernel_ms_per_step": {k: round(v[1] / args.steps, 3) for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])[:12]}}))
