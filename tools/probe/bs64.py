"""Headline model at batch 64 (BASELINE config 2): step time and per-launch table (development probe)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import fdet_amd
from fdet_amd import hotpath as hp
from fdet_amd.hostinfo import limit_host_threads
from fdet_amd.convstack import KernelTimer
from fdet_amd.datasets.synthetic import synthetic_boxes
from fdet_amd.models import ModelMeta
from fdet_amd.models.PoolResnet import PoolResnet
limit_host_threads()
dev = torch.device("cuda", 0)
for B in (64, 32, 256):
    model = PoolResnet(filters=64, input_shape=(3, 480, 480), num_of_patches=10).to(dev).train()
    mm = ModelMeta(model=model, lr=1e-4); mm.configure_optimizers()
    x = torch.rand(B, 3, 480, 480).to(dev)
    y = hp.encode_targets(synthetic_boxes(B, 480, seed=4), (480, 480), 10, device=dev)
    for _ in range(3):
        mm.fused_train_step(x, y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        mm.fused_train_step(x, y)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(f"bs {B}: {dt*1e3:.3f} ms/step, {B/dt:.0f} img/s")
    t = KernelTimer(); model.engine.timer = t
    mm.fused_train_step(x, y)
    model.engine.timer = None
    for k, v in sorted(t.summary().items(), key=lambda kv: -kv[1][1])[:12]:
        print(f"   {v[1]:7.3f} ms {k}")
    del model, mm, x, y
