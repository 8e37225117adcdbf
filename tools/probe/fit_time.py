"""Where does a trainer.fit() step spend its host time?  (development probe)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import fdet_amd
from fdet_amd import hotpath as hp
from fdet_amd.models import ModelMeta
from fdet_amd.models.PoolResnet import PoolResnet
from fdet_amd.datasets.feed import U8BatchFeeder
from fdet_amd.datasets.synthetic import synthetic_boxes

B, size, S = 256, 480, 10
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = PoolResnet(filters=64, input_shape=(3, size, size), num_of_patches=S).to(dev).train()
mm = ModelMeta(model=model, lr=1e-4); mm.configure_optimizers()
g = torch.Generator().manual_seed(1)
frames = torch.randint(0, 256, (B, 3, size, size), dtype=torch.uint8, generator=g)
y = hp.encode_targets(synthetic_boxes(B, size, seed=5), (size, size), S, device=dev).cpu()
pinned = [frames.pin_memory(), frames.roll(1, 0).pin_memory()]
t0 = time.perf_counter(); v = [p.is_pinned() for p in pinned * 5]; print("is_pinned x10", time.perf_counter() - t0, v[:2], frames.is_pinned())
fd = U8BatchFeeder((B, 3, size, size), (size, size), dev, target_shape=(B, 5, S, S), depth=3)
T = {"submit": 0.0, "get": 0.0, "step": 0.0, "release": 0.0}
fd.submit(pinned[0], y)
n = 12
for i in range(n):
    if i == 2:
        torch.cuda.synchronize(); T = {k: 0.0 for k in T}; w0 = time.perf_counter()
    a = time.perf_counter(); x_d, y_d, tok = fd.get(); b = time.perf_counter()
    mm.fused_train_step(x_d, y_d, with_metrics=True); c = time.perf_counter()
    fd.release(tok); d = time.perf_counter()
    fd.submit(pinned[(i + 1) % 2], y); e = time.perf_counter()
    T["get"] += b - a; T["step"] += c - b; T["release"] += d - c; T["submit"] += e - d
torch.cuda.synchronize(); wall = time.perf_counter() - w0
print("metrics on ", {k: round(v / (n - 2) * 1e3, 3) for k, v in T.items()}, "wall ms/step", round(wall / (n - 2) * 1e3, 3))
for i in range(n):
    if i == 2:
        torch.cuda.synchronize(); T = {k: 0.0 for k in T}; w0 = time.perf_counter()
    a = time.perf_counter(); x_d, y_d, tok = fd.get(); b = time.perf_counter()
    mm.fused_train_step(x_d, y_d, with_metrics=False); c = time.perf_counter()
    fd.release(tok); d = time.perf_counter()
    fd.submit(pinned[(i + 1) % 2], y); e = time.perf_counter()
    T["get"] += b - a; T["step"] += c - b; T["release"] += d - c; T["submit"] += e - d
torch.cuda.synchronize(); wall = time.perf_counter() - w0
print("metrics off", {k: round(v / (n - 2) * 1e3, 3) for k, v in T.items()}, "wall ms/step", round(wall / (n - 2) * 1e3, 3))
# host time of the pieces of one step, GPU idle in between
x_d, y_d, tok = fd.get()
for name, fn in (("step no metrics", lambda: mm.fused_train_step(x_d, y_d)), ("step metrics", lambda: mm.fused_train_step(x_d, y_d, with_metrics=True)),
                 ("metrics only", lambda: mm._metrics(mm.model(x_d[:1]).detach().expand(B, -1, -1, -1).contiguous(), y_d))):
    for r in range(3):
        torch.cuda.synchronize(); a = time.perf_counter(); fn(); b = time.perf_counter(); torch.cuda.synchronize(); c = time.perf_counter()
    print(name, "host ms", round((b - a) * 1e3, 3), "total ms", round((c - a) * 1e3, 3))
fd.release(tok)
# the same with the feeder's own pinned slots filled in place (bench feed_inclusive)
for i in range(3):
    x_d, y_d, tok = fd.get() if fd._inflight else (None, None, None)
    if tok is not None: fd.release(tok)
torch.cuda.synchronize()
