#!/usr/bin/env python3
"""Reference rates of plain torch kernels on this GPU (not product code): fill (write only), copy (read + write), sum (read only)."""
import json, torch
n = 512 * 1024 * 1024
x = torch.empty(n, dtype=torch.bfloat16, device="cuda"); y = torch.empty_like(x)
def t(f, reps=10):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
out = {}
ms = t(lambda: x.fill_(1.0)); out["fill_write_TBps"] = round(n * 2 / ms / 1e9, 2)
ms = t(lambda: y.copy_(x)); out["copy_read_plus_write_TBps"] = round(2 * n * 2 / ms / 1e9, 2)
xf = x.view(torch.float32)
ms = t(lambda: xf.sum()); out["sum_read_TBps"] = round(n * 2 / ms / 1e9, 2)
print(json.dumps(out))
