#!/usr/bin/env python3
"""Runs each conv primitive a few times at the bench shapes (target of rocprofv3 --pmc runs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fdet_amd
from fdet_amd import hotpath as hp
N, C = 256, 64
for H in (60, 15):
    x = torch.randn(N, C, H, H, device="cuda"); w = torch.randn(C, C, 3, 3, device="cuda") * 0.05; b = torch.randn(C, device="cuda")
    dz = torch.randn(N, C, H, H, device="cuda"); y = torch.empty_like(x)
    nf, nb = hp.packed_sizes(C, C)
    wf = torch.empty(nf, device="cuda"); wb = torch.empty(nb, device="cuda"); hp.pack_conv3x3_weights(w, wf, wb)
    ws = torch.empty(hp.conv3x3_wgrad_ws_bytes(N, C, C, H, H) // 4, device="cuda")
    dW = torch.empty_like(w); db = torch.empty_like(b)
    for _ in range(3):
        hp.conv3x3_fwd(x, wf, b, C, y_full=y)
        hp.conv3x3_wgrad(x, dz, dW, db, ws)
    torch.cuda.synchronize()
