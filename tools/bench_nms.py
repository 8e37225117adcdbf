#!/usr/bin/env python3
"""Batched greedy NMS throughput at config 5's candidate counts (K = 1024 / 4096 boxes per image,
scores U(0,1), centres uniform, sizes log-uniform 8..128 px, seed 2), 256 images per launch, beside the CPU
restatement (oracle.nms, one image at a time) on a bounded sample.   python tools/bench_nms.py"""
import json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fdet_amd
from fdet_amd import hotpath as hp
import oracle as O


def candidates(B, K, seed=2, size=480):
    g = torch.Generator().manual_seed(seed)
    c = torch.rand(B, K, 2, generator=g) * size
    wh = torch.exp(torch.rand(B, K, 2, generator=g) * (math.log(128.0) - math.log(8.0)) + math.log(8.0))
    return torch.cat([c - wh / 2, c + wh / 2], 2).round(), torch.rand(B, K, generator=g)


out = {}
for K in (1024, 4096):
    B = 256
    boxes, scores = candidates(B, K)
    bd, sd = boxes.cuda(), scores.cuda()
    cnt = torch.full((B,), K, dtype=torch.int32, device="cuda")
    for _ in range(3): keep, kc = hp.nms_batched(bd, sd, cnt, 0.5)
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): keep, kc = hp.nms_batched(bd, sd, cnt, 0.5)
    e.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(e) / 10
    t0 = time.time(); ref = [O.nms(boxes[b], scores[b], 0.5) for b in range(4)]; cpu_ms = (time.time() - t0) / 4 * 1e3
    ok = all(keep[b, : int(kc[b])].cpu().tolist() == ref[b].tolist() for b in range(4))
    out[f"K={K}"] = {"images_per_launch": B, "ms_per_launch": round(ms, 3), "images_per_s": round(B / ms * 1e3), "kept_mean": float(kc.float().mean()),
                     "cpu_oracle_ms_per_image": round(cpu_ms, 2), "parity_first_4_images": ok}
print(json.dumps(out))
