"""Why is the 32->16 data gradient at 240x240 slow? (development probe)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import fdet_amd
from fdet_amd import hotpath as hp
N, H = 64, 240
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for (ci, co) in ((16, 32), (32, 32)):
    w = torch.randn(co, ci, 3, 3, device="cuda") * 0.1
    nf, nb = hp.packed_sizes(co, ci)
    wf = torch.empty(nf, device="cuda"); wb = torch.empty(nb, device="cuda")
    hp.pack_conv3x3_weights(w, wf, wb, x3=True)
    dz = torch.randn(N, co, H, H, device="cuda")
    dx = torch.empty(N, ci, H, H, device="cuda")
    add = torch.randn(N, ci, H, H, device="cuda")
    act = torch.randn(N, ci, H, H, device="cuda")
    print(ci, co, "dgrad add ", t(lambda: hp.conv3x3_dgrad(dz, wb, ci, dx, add=add, slope=0.2, x3=True)))
    print(ci, co, "dgrad act ", t(lambda: hp.conv3x3_dgrad(dz, wb, ci, dx, act=act, slope=0.2, x3=True)))
    print(ci, co, "dgrad none", t(lambda: hp.conv3x3_dgrad(dz, wb, ci, dx, slope=0.2, x3=True)))
    x = torch.randn(N, ci, H, H, device="cuda"); y = torch.empty(N, co, H, H, device="cuda"); b = torch.zeros(co, device="cuda")
    print(ci, co, "fwd       ", t(lambda: hp.conv3x3_fwd(x, wf, b, co, y_full=y, slope=0.2, x3=True)))
