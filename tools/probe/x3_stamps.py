#!/usr/bin/env python3
"""Where does a workgroup of k_conv3x3_x3 spend its cycles?  (development aid)
Loads the -DFDET_X3_STAMPS build of fdet_conv3x3_x3.hip (tools/probe/libx3probe.so), runs the
forward / data-gradient conv at the bench shapes and prints the median cycles between stamps."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import fdet_amd
from fdet_amd import hotpath as hp

L = ctypes.CDLL(os.path.join(ROOT, "tools", "probe", "libx3probe.so"))
P, I, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
L.fdet_conv3x3_fwd_bf16x3.argtypes = [P, P, P, P, P, P, P, I, I, I, I, I, I, F, P]
L.fdet_conv3x3_dgrad_bf16x3.argtypes = [P, P, P, P, P, I, I, I, I, I, F, P]
L.fdet_x3_probe_set.argtypes = [P]
N, C = 256, 64
names = ["prologue", "chunk0", "chunk1", "chunk2", "chunk3", "epilogue0", "tile1-chunks", "epilogue1"]
for H in (60,):
    x = torch.randn(N, C, H, H, device="cuda"); w = torch.randn(C, C, 3, 3, device="cuda") * 0.05; b = torch.randn(C, device="cuda")
    y = torch.empty_like(x); nf, nb = hp.packed_sizes(C, C)
    wf = torch.empty(nf, device="cuda"); wb = torch.empty(nb, device="cuda"); hp.pack_conv3x3_weights(w, wf, wb, x3=True)
    st = torch.cuda.current_stream().cuda_stream
    for cfg in ("8,2", "4,4", "8,1"):
        os.environ["FDET_CONV_TILE"] = cfg
        stamps = torch.zeros(8192 * 32, dtype=torch.int64, device="cuda")
        for mode in ("fwd", "dgrad"):
            def run():
                if mode == "fwd":
                    rc = L.fdet_conv3x3_fwd_bf16x3(x.data_ptr(), wf.data_ptr(), b.data_ptr(), y.data_ptr(), None, None, None, N, C, C, H, H, 1, 0.2, st)
                else:
                    rc = L.fdet_conv3x3_dgrad_bf16x3(x.data_ptr(), wb.data_ptr(), y.data_ptr(), None, y.data_ptr(), N, C, C, H, H, 0.2, st)
                assert rc == 0
            L.fdet_x3_probe_set(None)
            for _ in range(3):
                run()
            L.fdet_x3_probe_set(stamps.data_ptr())
            stamps.zero_()
            run()
            torch.cuda.synchronize()
            L.fdet_x3_probe_set(None)
            s = stamps.view(-1, 32).cpu()
            s = s[s[:, 0] != 0]
            d = (s[:, 1:9] - s[:, 0:8]).double()
            tot = (s[:, 12] - s[:, 0]).double()
            print(f"   tiles per WG: min {int(s[:, 13].min())} max {int(s[:, 13].max())}; drain {float((s[:, 12] - s[:, 11]).double().median()):.0f}")
            real = (s[:, 14] - s[:, 15]).double() * 10.0     # ns (100 MHz)
            clk = float((tot / real).median())
            span_us = float(s[:, 14].max() - s[:, 15].min()) / 100.0
            med = d.median(0).values
            print(f"H={H} cfg={cfg} {mode}: {s.shape[0]} WGs, total median {float(tot.median()):.0f} cyc, clock {clk:.2f} GHz, kernel span {span_us:.1f} us")
            print("   " + "  ".join(f"{n}={int(v)}" for n, v in zip(names, med)))
            s2 = s[s[:, 16] != 0]
            if s2.shape[0]:
                d2 = (s2[:, 17:22] - s2[:, 16:21]).double().median(0).values
                print("   tile1 chunk1: " + "  ".join(f"{n}={int(v)}" for n, v in zip(["load-issue", "taps0-2", "taps3-5", "taps6-8", "barrier"], d2)))
os.environ.pop("FDET_CONV_TILE", None)
