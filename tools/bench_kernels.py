#!/usr/bin/env python3
"""Per-kernel timing of the conv primitives at the bench shapes (development aid).
   python tools/bench_kernels.py [--batch 256] [--reps 10]"""
import argparse, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fdet_amd
from fdet_amd import hotpath as hp

ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=256); ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--filters", type=int, default=64)
ap.add_argument("--x3", type=int, default=1)
args = ap.parse_args()
N, C = args.batch, args.filters


def timeit(fn, reps=args.reps):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

peak = os.path.join(ROOT, "tools", "probe", "libpeak.so")
if os.path.exists(peak):
    L = ctypes.CDLL(peak)
    out = torch.empty(4096 * 256, device="cuda")
    for blocks, iters in ((2048, 2000), (512, 8000)):
        ms = timeit(lambda: L.probe_peak(ctypes.c_void_p(out.data_ptr()), blocks, iters, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), 3)
        fl = blocks * 4 * iters * 8 * 32 * 32 * 2 * 2
        print(f"mfma_f32_32x32x2 peak probe blocks={blocks}: {ms:.3f} ms -> {fl/ms/1e9:.1f} TFLOP/s")
    rnd = torch.randn(4096, device="cuda") * 0.01
    ms = timeit(lambda: L.probe_peak_rand(ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(rnd.data_ptr()), 2048, 2000, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), 5)
    ms2 = timeit(lambda: L.probe_peak_lds(ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(rnd.data_ptr()), 2048, 2000, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), 5)
    print(f"mfma_f32_32x32x2 + LDS operand reads (conv pattern, random): {ms2:.3f} ms -> {2048*4*2000*8*32*32*2*2/ms2/1e9:.1f} TFLOP/s")
    print(f"mfma_f32_32x32x2 peak probe RANDOM operands: {ms:.3f} ms -> {2048*4*2000*8*32*32*2*2/ms/1e9:.1f} TFLOP/s")

for H in (60, 30, 15):
    x = torch.randn(N, C, H, H, device="cuda"); w = torch.randn(C, C, 3, 3, device="cuda") * 0.05; b = torch.randn(C, device="cuda")
    dz = torch.randn(N, C, H, H, device="cuda"); y = torch.empty_like(x); y2 = torch.empty_like(x)
    nf, nb = hp.packed_sizes(C, C)
    wf = torch.empty(nf, device="cuda"); wb = torch.empty(nb, device="cuda"); hp.pack_conv3x3_weights(w, wf, wb, x3=bool(args.x3))
    ws = torch.empty(hp.conv3x3_wgrad_ws_bytes(N, C, C, H, H) // 4, device="cuda")
    dW = torch.empty_like(w); db = torch.empty_like(b)
    fl = 2.0 * N * C * C * 9 * H * H
    sc = torch.ones(N, C, device="cuda")
    t1 = timeit(lambda: hp.conv3x3_fwd(x, wf, b, C, y_full=y, x3=bool(args.x3)))
    t1b = timeit(lambda: hp.conv3x3_fwd(x, wf, b, C, y_full=y, skip=x, drop_scale=sc, y_out=y2, x3=bool(args.x3)))
    t2 = timeit(lambda: hp.conv3x3_dgrad(dz, wb, C, y, act=x, x3=bool(args.x3)))
    t2b = timeit(lambda: hp.conv3x3_dgrad(dz, wb, C, y, add=x, x3=bool(args.x3)))
    t3 = timeit(lambda: hp.conv3x3_wgrad(x, dz, dW, db, ws, x3=bool(args.x3)))
    print(f"{H}x{H}: fwd {t1:.3f} ms ({fl/t1/1e9:.1f} TF) | fwd+tail {t1b:.3f} ({fl/t1b/1e9:.1f}) | dgrad+act {t2:.3f} ({fl/t2/1e9:.1f}) | "
          f"dgrad+add {t2b:.3f} ({fl/t2b/1e9:.1f}) | wgrad {t3:.3f} ({fl/t3/1e9:.1f} TF)   ideal@157TF {fl/157.3e9:.3f} ms")
