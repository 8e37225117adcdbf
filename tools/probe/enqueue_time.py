"""Host-side cost of one training step: time to ENQUEUE K steps (no synchronisation inside) against the time until the GPU
has finished them.  If the first approaches the second, the step is launch-bound and faster kernels stop showing."""
import os, sys, time, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from fdet_amd.models import ModelMeta
from fdet_amd.models.PoolResnet import PoolResnet
B, size, S = 256, 480, 10
model = PoolResnet(64, (3, size, size), S).cuda().train()
mm = ModelMeta(model=model, lr=1e-4); mm.configure_optimizers()
x, y, _ = bench.synth_batch(B, size, S, 0, "cuda")
for _ in range(5): mm.fused_train_step(x, y)
torch.cuda.synchronize()
K = 30
t0 = time.perf_counter()
for _ in range(K): mm.fused_train_step(x, y)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(json.dumps({"enqueue_ms_per_step": round((t1 - t0) / K * 1e3, 3), "total_ms_per_step": round((t2 - t0) / K * 1e3, 3)}))
