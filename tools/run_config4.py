#!/usr/bin/env python3
"""BASELINE.json config 4 on ONE GPU: SSD (filters 16) at 3x480x480, 4774 priors, hard-negative ratio 10;
fused training steps (forward + ssd_loss + backward + Adam), reported as imgs/s, next to the CPU oracle
(autograd over torch ops) on a bounded sample.   python tools/run_config4.py [--batch 64] [--steps 5]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fdet_amd
from fdet_amd import hotpath as hp
from fdet_amd.models.SSD import SSD
from fdet_amd.models.ModelMetaSSD import ModelMetaSSD
import oracle as O
from oracle import ssd_model_oracle as SM

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--cpu-batch", type=int, default=4)
args = ap.parse_args()
B, SIZE, FIL = args.batch, 480, 16
torch.manual_seed(0)
model = SSD(filters=FIL, input_shape=(3, SIZE, SIZE)).cuda().train()
mm = ModelMetaSSD(model=model, lr=1e-4); mm.configure_optimizers()
g = torch.Generator().manual_seed(1)
x = torch.rand(B, 3, SIZE, SIZE, generator=g).cuda()
boxes = O.synthetic_boxes(B, SIZE, seed=2, max_faces=6)
y = hp.ssd_encode_targets(boxes, (SIZE, SIZE))
for _ in range(2):
    loss, _ = mm.fused_train_step(x, y)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(args.steps):
    loss, yh = mm.fused_train_step(x, y)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / args.steps
# CPU oracle (bounded sample)
nb = args.cpu_batch
torch.set_num_threads(min(16, os.cpu_count() or 1))
P = SM.init_params(FIL, 0)
xc, yc = x[:nb].cpu(), y[:nb].cpu()
masks = SM.make_dropout_masks(FIL, nb, seed=1)
SM.loss_and_grads(FIL, P, xc, yc, masks)
t0 = time.perf_counter(); SM.loss_and_grads(FIL, P, xc, yc, masks); ct = time.perf_counter() - t0
print(json.dumps({"config": f"SSD filters 16, 3x480x480, 4774 priors, bs {B}, 1 GPU: fwd + ssd_loss (hard-neg ratio 10) + bwd + Adam",
                  "ms_per_step": round(dt * 1e3, 2), "imgs_per_s": round(B / dt, 1), "loss": float(loss), "finite": bool(torch.isfinite(yh).all()),
                  "cpu_oracle": {"imgs_per_s": round(nb / ct, 2), "what": f"forward + ssd_loss + autograd backward of {nb} images, torch CPU fp32, {torch.get_num_threads()} threads (no optimiser step)"}}))
