#!/usr/bin/env python3
"""Runs tools/probe/mfma_shapes.hip: PFLOP/s of the same bf16 work as 32x32x16 and as 16x16x32 MFMAs (random operands,
registers / LDS-fed), one wave per SIMD on every CU.  Per iteration a wave does a 64x64 block at K=32: 2*64*64*32 flops."""
import ctypes, json, os, subprocess, sys
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(HERE, "libmfmashapes.so")
if not os.path.exists(so):
    subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", os.path.join(HERE, "mfma_shapes.hip"), "-o", so], check=True)
L = ctypes.CDLL(so)
L.probe_shape.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
blocks, iters = 256 * 4, 20000
out = torch.empty(blocks * 256, device="cuda")
inp = (torch.randn(2048 * 8, device="cuda")).to(torch.bfloat16)
st = torch.cuda.current_stream().cuda_stream
res = {}
for shape in (32, 16):
    for lds in (0, 1):
        for _ in range(2):
            L.probe_shape(out.data_ptr(), inp.data_ptr(), blocks, iters, shape, lds, st)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            L.probe_shape(out.data_ptr(), inp.data_ptr(), blocks, iters, shape, lds, st)
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 3
        flops = blocks * 4 * iters * 2.0 * 64 * 64 * 32
        res[f"{shape}x{shape}{'_lds' if lds else '_reg'}"] = {"ms": round(ms, 3), "pflops": round(flops / ms / 1e12, 3)}
print(json.dumps(res))
