"""MobileNetV3 training step under rocprofv3 (development probe)."""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import fdet_amd
from fdet_amd import hotpath as hp
from fdet_amd.models import ModelMeta
from fdet_amd.models.MobilenetV3Backbone import MobilenetV3Backbone
from fdet_amd.datasets.synthetic import synthetic_boxes
dev = torch.device("cuda", 0)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    net = MobilenetV3Backbone(64, (3, 480, 480), 15, pretrained=False).to(dev).train()
mm = ModelMeta(model=net, lr=1e-4)
(opt,), _ = mm.configure_optimizers()
B = 32
x = torch.rand(B, 3, 480, 480).to(dev)
bx = synthetic_boxes(B, 480, seed=9)
y = hp.encode_targets(bx, (480, 480), 15, device=dev)
for _ in range(3):
    o = mm.training_step((x, y, bx), 0)
    opt.zero_grad(); o["loss"].backward(); opt.step()
torch.cuda.synchronize()
print("loss", float(o["loss"]))
import time
t0 = time.perf_counter()
for _ in range(10):
    o = mm.training_step((x, y, bx), 0)
    opt.zero_grad(); o["loss"].backward(); opt.step()
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) * 100)
for _ in range(3):
    mm.fused_train_step(x, y)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    mm.fused_train_step(x, y)
torch.cuda.synchronize()
print("fused ms/step", (time.perf_counter() - t0) * 100)
