"""One of the side configs' training step for rocprofv3 (development probe): python cfg_step.py {3|4|128}"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import fdet_amd
from fdet_amd import hotpath as hp
from fdet_amd.hostinfo import limit_host_threads
from fdet_amd.datasets.synthetic import synthetic_boxes
from fdet_amd.models import ModelMeta
limit_host_threads()
which = sys.argv[1] if len(sys.argv) > 1 else "3"
dev = torch.device("cuda", 0)
torch.manual_seed(0)
if which == "3":
    from fdet_amd.models.Resnet import Resnet
    B, size, S = 32, 640, 20
    model = Resnet(filters=64, input_shape=(3, size, size), num_of_patches=S, num_of_residual_blocks=10).to(dev).train()
    mm = ModelMeta(model=model, lr=1e-4); mm.configure_optimizers()
    y = hp.encode_targets(synthetic_boxes(B, size, seed=4), (size, size), S, device=dev)
elif which == "128":
    from fdet_amd.models.PoolResnet import PoolResnet
    B, size, S = 256, 480, 10
    model = PoolResnet(filters=128, input_shape=(3, size, size), num_of_patches=S, num_of_residual_blocks=10).to(dev).train()
    mm = ModelMeta(model=model, lr=1e-4); mm.configure_optimizers()
    y = hp.encode_targets(synthetic_boxes(B, size, seed=4), (size, size), S, device=dev)
else:
    from fdet_amd.models.SSD import SSD
    from fdet_amd.models.ModelMetaSSD import ModelMetaSSD
    B, size = 64, 480
    model = SSD(filters=16, input_shape=(3, size, size)).to(dev).train()
    mm = ModelMetaSSD(model=model, lr=1e-4); mm.configure_optimizers()
    y = hp.ssd_encode_targets(synthetic_boxes(B, size, seed=2), (size, size), device=dev)
x = torch.rand(B, 3, size, size).to(dev)
for _ in range(2):
    mm.fused_train_step(x, y)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    mm.fused_train_step(x, y)
torch.cuda.synchronize()
print("config", which, "ms/step", (time.perf_counter() - t0) * 200)
