#!/usr/bin/env python3
"""Sweep (MT,NT) tile configs of the conv3x3 kernel at the bench shapes (development aid)."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch, fdet_amd
    from fdet_amd import hotpath as hp
    N, C = int(sys.argv[2]), 64
    res = {}
    for H in (60, 30, 15):
        x = torch.randn(N, C, H, H, device="cuda"); w = torch.randn(C, C, 3, 3, device="cuda") * 0.05; b = torch.randn(C, device="cuda")
        y = torch.empty_like(x); nf, nb = hp.packed_sizes(C, C)
        wf = torch.empty(nf, device="cuda"); hp.pack_conv3x3_weights(w, wf, None)
        try:
            for _ in range(2): hp.conv3x3_fwd(x, wf, b, C, y_full=y)
            torch.cuda.synchronize()
            a = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True); a.record()
            for _ in range(10): hp.conv3x3_fwd(x, wf, b, C, y_full=y)
            e.record(); torch.cuda.synchronize()
            res[H] = a.elapsed_time(e) / 10
        except Exception as ex:
            res[H] = None
    print(json.dumps(res))
else:
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    cfgs = [(2, 4, st) for st in (0, 2, 5, 10, 20)] + [(2, 2, st) for st in (0, 2, 5, 10)] + [(1, 2, st) for st in (0, 1, 2, 4, 8)] + [(2, 1, st) for st in (0, 2, 4)]
    for mt, nt, dbg in cfgs:
        if True:
            env = dict(os.environ, FDET_CONV_TILE=f"{mt},{nt}", FDET_CONV_STAGGER=str(dbg))
            r = subprocess.run([sys.executable, __file__, "child", str(N)], env=env, capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            print(f"MT={mt} NT={nt} stagger={dbg}:", line[-1] if line else r.stderr[-300:], flush=True)
