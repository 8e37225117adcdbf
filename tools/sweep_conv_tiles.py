#!/usr/bin/env python3
"""Sweep (MT,NT,SB) tile configs of the bf16x3 conv3x3 kernel at the bench shapes (development aid).
One process: FDET_CONV_TILE is read by the library at every call.  Each config is also checked
against the fp32 torch conv (max abs error printed)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, fdet_amd
from fdet_amd import hotpath as hp

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
C = 64
cfgs = [(2, 2, 0), (2, 1, 0), (1, 2, 0), (1, 1, 0)]


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True); a.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / reps


for H in (60, 30, 15):
    x = torch.randn(N, C, H, H, device="cuda"); w = torch.randn(C, C, 3, 3, device="cuda") * 0.05; b = torch.randn(C, device="cuda")
    dz = torch.randn(N, C, H, H, device="cuda")
    y = torch.empty_like(x); y2 = torch.empty_like(x); nf, nb = hp.packed_sizes(C, C)
    wf = torch.empty(nf, device="cuda"); wb = torch.empty(nb, device="cuda"); hp.pack_conv3x3_weights(w, wf, wb, x3=True)
    ref = torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(x[:8], w, b, padding=1), 0.2)
    for mt, nt, sb in cfgs:
        os.environ["FDET_SB_TILE"] = f"{mt},{nt}"
        os.environ["FDET_CONV_STAGGER"] = str(sb)
        try:
            t1 = timeit(lambda: hp.conv3x3_fwd(x, wf, b, C, y_full=y, x3=True))
            err = float((y[:8] - ref).abs().max())
            t2 = timeit(lambda: hp.conv3x3_dgrad(dz, wb, C, y2, act=x, x3=True))
            print(f"H={H} MT={mt} NT={nt} stagger={sb}: fwd {t1*1e3:.1f} us  dgrad+act {t2*1e3:.1f} us  maxerr {err:.2e}", flush=True)
        except Exception as ex:
            print(f"H={H} MT={mt} NT={nt} stagger={sb}: {str(ex)[:120]}", flush=True)
os.environ.pop("FDET_CONV_TILE", None); os.environ.pop("FDET_CONV_STAGGER", None)
