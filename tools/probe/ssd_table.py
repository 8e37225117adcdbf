"""Per-launch table of one SSD (config 4) training step (development probe)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import fdet_amd
from fdet_amd import hotpath as hp
from fdet_amd.hostinfo import limit_host_threads
from fdet_amd.convstack import KernelTimer
from fdet_amd.datasets.synthetic import synthetic_boxes
from fdet_amd.models.SSD import SSD
from fdet_amd.models.ModelMetaSSD import ModelMetaSSD
limit_host_threads()
dev = torch.device("cuda", 0)
B, size = 64, 480
model = SSD(filters=16, input_shape=(3, size, size)).to(dev).train()
mm = ModelMetaSSD(model=model, lr=1e-4); mm.configure_optimizers()
y = hp.ssd_encode_targets(synthetic_boxes(B, size, seed=2), (size, size), device=dev)
x = torch.rand(B, 3, size, size).to(dev)
for _ in range(2):
    mm.fused_train_step(x, y)
t = KernelTimer(); model.engine.timer = t
mm.fused_train_step(x, y)
model.engine.timer = None
tot = 0.0
for k, (n, ms, fl, nb) in sorted(t.summary().items(), key=lambda kv: -kv[1][1]):
    tot += ms
    print(f"{ms:8.3f} ms  {n:3d} x  {k}")
print("sum", tot)
