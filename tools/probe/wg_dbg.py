#!/usr/bin/env python3
"""Times the batched bf16x3 weight gradient at the bench shapes (development aid; with a
-DFDET_WG_DBG build, env FDET_WG_DBG=<bits> switches parts of the pipelined kernel off)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import fdet_amd
from fdet_amd import hotpath as hp

N, C = 256, 64
for H, L in ((60, 2), (30, 4), (15, 16)):
    xs = [torch.randn(N, C, H, H, device="cuda") for _ in range(L)]
    dzs = [torch.randn(N, C, H, H, device="cuda") for _ in range(L)]
    dWs = [torch.empty(C, C, 3, 3, device="cuda") for _ in range(L)]
    dbs = [torch.empty(C, device="cuda") for _ in range(L)]
    ws = torch.empty(hp.conv3x3_wgrad_batched_ws_bytes(L, N, C, C, H, H) // 4, device="cuda")
    for _ in range(3): hp.conv3x3_wgrad_batched(xs, dzs, dWs, dbs, ws)
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): hp.conv3x3_wgrad_batched(xs, dzs, dWs, dbs, ws)
    b.record(); torch.cuda.synchronize()
    print(f"DBG={os.environ.get('FDET_WG_DBG','0')} {H}x{H} L={L}: {a.elapsed_time(b)/10:.3f} ms")
