#!/usr/bin/env python3
"""Where do the phases of the ping-pong conv kernel (fdet_conv3x3_x3_pp.hip) spend their cycles?  (development aid)
Builds tools/probe/libppprobe.so from the kernel source with -DFDET_PP_STAMPS, runs the forward conv at the bench
shape and prints, per wave of a few workgroups: the SIMD it sits on, and the median cycles of its MFMA segments,
its memory segments, and the time it then waits at the barrier."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
SRC = os.path.join(ROOT, "pytorch-face-detection-from-scratch_amd", "csrc")
VARIANT = os.environ.get("PP_VARIANT", "")            # e.g. "NOP=0,PRIO=2" -> -DFDET_PP_NOP=0 -DFDET_PP_PRIO=2
SO = os.path.join(ROOT, "tools", "probe", "libppprobe" + VARIANT.replace("=", "").replace(",", "_") + ".so")
if "--build" in sys.argv or not os.path.exists(SO):
    defs = ["-DFDET_PP_" + d for d in VARIANT.split(",") if d]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-ffp-contract=off",
                           "-DFDET_PP_STAMPS", *defs, "-I" + os.path.join(ROOT, "include"), os.path.join(SRC, "fdet_conv3x3_x3_pp.hip"), "-o", SO])
    if "--build" in sys.argv:
        sys.exit(0)
import torch
import fdet_amd
from fdet_amd import hotpath as hp

L = ctypes.CDLL(SO)
P, I = ctypes.c_void_p, ctypes.c_int
L.fdet_pp_probe_conv.argtypes = [P, P, P, P, P, I, I, I, I, I, P]
L.fdet_pp_probe_set.argtypes = [P]
N, C = 256, 64
for H in ((60,) if VARIANT else (60, 30)):
    x = torch.randn(N, C, H, H, device="cuda"); w = torch.randn(C, C, 3, 3, device="cuda") * 0.05; b = torch.randn(C, device="cuda")
    y = torch.empty_like(x); nf, nb = hp.packed_sizes(C, C)
    wf = torch.empty(nf, device="cuda"); wb = torch.empty(nb, device="cuda"); hp.pack_conv3x3_weights(w, wf, wb, x3=True)
    st = torch.cuda.current_stream().cuda_stream
    for dgrad in (0, 1):
        stamps = torch.zeros(32 * 8 * 256, dtype=torch.int64, device="cuda")
        def run():
            assert L.fdet_pp_probe_conv(x.data_ptr(), (wb if dgrad else wf).data_ptr(), b.data_ptr(), y.data_ptr(), x.data_ptr(), N, C, H, H, dgrad, st) == 0
        L.fdet_pp_probe_set(None)
        for _ in range(3):
            run()
        L.fdet_pp_probe_set(stamps.data_ptr())
        run()
        torch.cuda.synchronize()
        L.fdet_pp_probe_set(None)
        s = stamps.view(32, 8, 256).cpu()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(10):
            run()
        ev1.record(); torch.cuda.synchronize()
        print(f"=== H={H} {'dgrad(act)' if dgrad else 'fwd(full)'}  kernel time {ev0.elapsed_time(ev1) * 100:.1f} us (stagger env {os.environ.get('FDET_PP_STAGGER')})")
        for wg in ((0,) if VARIANT else (0, 1, 9)):
            nph = int(s[wg, 0, 1])
            print(f" WG {wg}: {nph} phases; total {int(s[wg, 0, 3 + 2 * (nph - 1)] - s[wg, 0, 2])} cycles = {float(s[wg, 0, 3 + 2 * (nph - 1)] - s[wg, 0, 2]) / nph:.0f} per phase")
            for wv in (0, 4):
                f = s[wg, wv, 200:204].double()
                print(f"   wave {wv} second epilogue segment: epi loads issue {f[1]-f[0]:.0f}  LDS writes of next chunk {f[2]-f[1]:.0f}  transposes+math+stores {f[3]-f[2]:.0f}")
            for wv in range(8):
                hw = int(s[wg, wv, 0]); simd = (hw >> 4) & 3; cu = (hw >> 8) & 15
                st_ = s[wg, wv, 2:2 + 2 * nph:2].double(); en = s[wg, wv, 3:3 + 2 * nph:2].double()
                work = en - st_
                wait = st_[1:] - en[:-1]
                grp = wv >> 2
                ph = torch.arange(nph)
                comp = ((ph - grp) >= 0) & (((ph - grp) & 1) == 0)
                mem = ~comp
                # boundary memory turns: kn % 4 == 0
                kn = (ph - grp + 1) >> 1
                bnd = mem & (kn % 4 == 0) & (kn >= 4)
                print(f"   wave {wv} simd {simd} cu {cu} hw {hw:#x}: MFMA seg {work[comp][1:-1].median():.0f}  mem seg {work[mem & ~bnd][2:-1].median():.0f}  "
                      f"mem+epilogue {work[bnd].median() if bnd.any() else float('nan'):.0f}  barrier wait after MFMA {wait[comp[:-1]][1:].median():.0f} after mem {wait[mem[:-1] & ~bnd[:-1]][2:].median():.0f} after epi {wait[bnd[:-1]].median() if bnd[:-1].any() else float('nan'):.0f}")
