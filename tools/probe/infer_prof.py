"""Batched inference (256 uint8 frames -> boxes) for rocprofv3 (development probe)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import fdet_amd
from fdet_amd.models.PoolResnet import PoolResnet
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = PoolResnet(filters=64, input_shape=(3, 480, 480), num_of_patches=10).to(dev).eval()
big = torch.randint(0, 256, (256, 3, 480, 480), dtype=torch.uint8).to(dev)
with torch.no_grad():
    for _ in range(3):
        model.non_max_suppression(model(model._preprocess(big)))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        model.non_max_suppression(model(model._preprocess(big)))
    torch.cuda.synchronize()
    print("ms per 256", (time.perf_counter() - t0) * 100)
