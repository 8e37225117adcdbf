"""PS weight gradient timing (development probe; FDET_LIB_PATH selects the build)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import fdet_amd
from fdet_amd import ps
for (N, H, L) in ((256, 60, 2), (256, 30, 2), (256, 15, 16), (192, 320, 1)):
    W = H
    if H == 320:
        N, W = 32, 320
    xs = [ps.PsTensor.from_f32(torch.randn(N, 64, H, W, device="cuda")) for _ in range(L)]
    zs = [ps.PsTensor.from_f32(torch.randn(N, 64, H, W, device="cuda")) for _ in range(L)]
    for z in zs:
        ps.halo_exchange(z, zero_only=True)
    dW = [torch.empty(64, 64, 3, 3, device="cuda") for _ in range(L)]; db = [torch.empty(64, device="cuda") for _ in range(L)]
    ws = torch.empty(ps.conv3x3_wgrad_ps_ws_bytes(L, N, 64, H, W) // 4, device="cuda")
    for p16 in (False, True):
        for _ in range(3):
            ps.conv3x3_wgrad_ps_batched(xs, zs, dW, db, ws, p16=p16)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            ps.conv3x3_wgrad_ps_batched(xs, zs, dW, db, ws, p16=p16)
        torch.cuda.synchronize()
        print(f"N={N} {H}x{W} L={L} p16={p16}: {(time.perf_counter() - t0) / 20 * 1e3:.4f} ms  checksum {float(dW[0].sum()):.6e}")
    del xs, zs
