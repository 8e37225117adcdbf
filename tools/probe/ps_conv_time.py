"""Timing of the PS conv kernel against the fp32-I/O bf16x3 kernels on the PoolResnet shapes (HIP events, interleaved
rounds in one process)."""
import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import fdet_amd
from fdet_amd import hotpath as hp, ps

def timeit(fns, rounds=10, inner=5):
    res = {k: [] for k in fns}
    for _ in range(rounds):
        for k, f in fns.items():
            a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(inner):
                f()
            b.record(); torch.cuda.synchronize()
            res[k].append(a.elapsed_time(b) / inner)
    return {k: (round(sorted(v)[len(v) // 2], 4), round(min(v), 4)) for k, v in res.items()}

out = {}
for (N, H) in ((256, 60), (256, 30)):
    C = 64
    g = torch.Generator().manual_seed(0)
    x = torch.randn(N, C, H, H, generator=g).cuda()
    w = (torch.randn(C, C, 3, 3, generator=g) * 0.05).cuda()
    b = torch.randn(C, generator=g).cuda()
    nf, nb = hp.packed_sizes(C, C)
    wf = torch.empty(nf, device="cuda"); wb = torch.empty(nb, device="cuda")
    hp.pack_conv3x3_weights(w, wf, wb, x3=True)
    y = torch.empty_like(x); act = torch.randn_like(x)
    xp = ps.PsTensor.from_f32(x); yp = ps.PsTensor(N, C, H, H, "cuda"); ap = ps.PsTensor.from_f32(act)
    fns = {
        "fwd_f32io": lambda: hp.conv3x3_fwd(x, wf, b, C, y_full=y, x3=True),
        "fwd_ps": lambda: ps.conv3x3_ps_fwd(xp, wf, b, yp),
        "dgrad_act_f32io": lambda: hp.conv3x3_dgrad(x, wb, C, y, act=act, x3=True),
        "dgrad_act_ps": lambda: ps.conv3x3_ps_dgrad_act(xp, wb, ap, yp),
    }
    for f in fns.values():
        f()
    torch.cuda.synchronize()
    r = timeit(fns)
    flops = 2.0 * N * C * C * 9 * H * H * 3
    out[f"{H}x{H}"] = {k: {"median_ms": v[0], "min_ms": v[1], "pflops_bf16_passes": round(flops / (v[0] * 1e-3) / 1e15, 3)} for k, v in r.items()}
print(json.dumps(out, indent=1))

# ---- weight gradient: PS kernel vs the register-staged pipeline, 2 layers at 60x60 / 30x30 as in the training step
wout = {}
for (N, H) in ((256, 60), (256, 30)):
    C, L = 64, 2
    g = torch.Generator().manual_seed(1)
    xs = [torch.randn(N, C, H, H, generator=g).cuda() for _ in range(L)]
    zs = [torch.randn(N, C, H, H, generator=g).cuda() for _ in range(L)]
    xp = [ps.PsTensor.from_f32(t) for t in xs]; zp = [ps.PsTensor.from_f32(t) for t in zs]
    dW = [torch.empty(C, C, 3, 3, device="cuda") for _ in range(L)]; db = [torch.empty(C, device="cuda") for _ in range(L)]
    ws_ps = torch.empty(ps.conv3x3_wgrad_ps_ws_bytes(L, N, C, H, H) // 4, device="cuda")
    ws_old = torch.empty(hp.conv3x3_wgrad_batched_ws_bytes(L, N, C, C, H, H) // 4, device="cuda")
    fns = {"wgrad_f32io": lambda: hp.conv3x3_wgrad_batched(xs, zs, dW, db, ws_old),
           "wgrad_ps": lambda: ps.conv3x3_wgrad_ps_batched(xp, zp, dW, db, ws_ps)}
    for f in fns.values():
        f()
    torch.cuda.synchronize()
    r = timeit(fns)
    flops = 2.0 * L * N * C * C * 9 * H * H * 3
    wout[f"{H}x{H}"] = {k: {"median_ms": v[0], "min_ms": v[1], "pflops_bf16_passes": round(flops / (v[0] * 1e-3) / 1e15, 3)} for k, v in r.items()}
print("WGRAD " + json.dumps(wout))
