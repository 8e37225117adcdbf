"""Why is the batched inference leg slow? (development probe)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import fdet_amd
from fdet_amd.models.PoolResnet import PoolResnet
from fdet_amd.convstack import KernelTimer
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = PoolResnet(filters=64, input_shape=(3, 480, 480), num_of_patches=10).to(dev).eval()
big = torch.randint(0, 256, (256, 3, 480, 480), dtype=torch.uint8).to(dev)
with torch.no_grad():
    for _ in range(2):
        model.non_max_suppression(model(model._preprocess(big)))
    torch.cuda.synchronize()
    for name, fn in (("preprocess", lambda: model._preprocess(big)), ("forward", lambda: model(model._preprocess(big))),
                     ("all", lambda: model.non_max_suppression(model(model._preprocess(big))))):
        t0 = time.perf_counter()
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        print(name, round((time.perf_counter() - t0) / 5 * 1e3, 3), "ms")
    t = KernelTimer(); model.engine.timer = t
    model(model._preprocess(big))
    model.engine.timer = None
    for k, v in sorted(t.summary().items(), key=lambda kv: -kv[1][1]):
        print(" ", k, round(v[1], 3))

# ---- the same after a precision16 training model lived in this process (bench.py's leg order)
from fdet_amd.models import ModelMeta
from fdet_amd import hotpath as hp
from fdet_amd.datasets.synthetic import synthetic_boxes


def p16_leg():
    m2 = PoolResnet(filters=64, input_shape=(3, 480, 480), num_of_patches=10).to(dev).train()
    m2.engine.set_precision("bf16")
    mm = ModelMeta(model=m2, lr=1e-4); mm.configure_optimizers()
    x = torch.rand(256, 3, 480, 480).to(dev)
    y = hp.encode_targets(synthetic_boxes(256, 480, seed=5), (480, 480), 10, device=dev)
    for _ in range(3):
        mm.fused_train_step(x, y)
    torch.cuda.synchronize()


p16_leg()
with torch.no_grad():
    for _ in range(2):
        model.non_max_suppression(model(model._preprocess(big)))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        model.non_max_suppression(model(model._preprocess(big)))
    torch.cuda.synchronize()
    print("all, after a p16 leg", round((time.perf_counter() - t0) / 5 * 1e3, 3), "ms")
    t = KernelTimer(); model.engine.timer = t
    model(model._preprocess(big))
    model.engine.timer = None
    for k, v in sorted(t.summary().items(), key=lambda kv: -kv[1][1]):
        print(" ", k, round(v[1], 3))
