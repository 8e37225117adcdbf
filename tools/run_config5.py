#!/usr/bin/env python3
"""BASELINE.json config 5 on ONE GPU: MobileNetV3-small backbone (models/MobilenetV3Backbone.py:11-60) forward in bf16 at
3x480x480, batch 256, with the reference archive's parameters (tests/golden/g13_mobilenet_weights.npz), layer by layer
against the HBM roofline, then batched greedy NMS over K = 1024 / 4096 candidates per image.

Per layer: algorithmic bytes = input activation read once + output written once (+ residual read), bf16; time from HIP
events around the launch on the stream it runs on; roof = 8 TB/s (MI355X_MICROARCH.md).  A layer is "hbm" bound when its
bytes/8 TB/s exceeds its flops/2.5 PFLOP/s (all of them are: 2-100 FLOP/B against a ridge of ~310).
Beside it the CPU oracle (fp32 torch restatement, PARITY UNPINNED) on a bounded sample.
   python tools/run_config5.py [--batch 256] [--reps 5] [--json out.json]"""
import argparse, json, math, os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import fdet_amd
from fdet_amd import hotpath as hp
from fdet_amd.models.MobilenetV3Backbone import MobilenetV3Backbone
import oracle as O
from oracle import mobilenet_oracle as MO

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--cpu-batch", type=int, default=4)
ap.add_argument("--json", default=None)
args = ap.parse_args()
B, SIZE = args.batch, 480
HBM, MFMA = 8.0e12, 2.5e15

z = np.load(os.path.join(ROOT, "tests", "golden", "g13_mobilenet_weights.npz"))
P = {k: torch.from_numpy(z[k]) for k in z.files}
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    net = MobilenetV3Backbone(64, (3, SIZE, SIZE), 15, pretrained=False)
net.load_state_dict(P)
net = net.cuda().eval()
x = torch.rand(B, 3, SIZE, SIZE, generator=torch.Generator().manual_seed(1)).cuda()
eng = net._packed_engine()


class Timer:
    def __init__(self):
        self.rows = {}

    def __call__(self, label, nbytes, flops):
        return _Span(self, label, nbytes, flops)


class _Span:
    def __init__(self, t, label, nbytes, flops):
        self.t, self.label, self.bytes, self.flops = t, label, nbytes, flops

    def __enter__(self):
        self.a = torch.cuda.Event(enable_timing=True); self.b = torch.cuda.Event(enable_timing=True)
        self.a.record()
        return self

    def __exit__(self, *exc):
        self.b.record()
        self.t.rows.setdefault(self.label, []).append((self.a, self.b, self.bytes, self.flops))
        return False


for _ in range(2):
    y = net(x)
torch.cuda.synchronize()
# whole forward, untimed layers (what a user gets)
a = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(args.reps):
    y = net(x)
e.record(); torch.cuda.synchronize()
fwd_ms = a.elapsed_time(e) / args.reps
# layer by layer
tm = Timer(); eng.timer = tm
for _ in range(args.reps):
    net(x)
torch.cuda.synchronize(); eng.timer = None
layers, tot_b, tot_roof = [], 0, 0.0
for label, spans in tm.rows.items():
    ms = sum(s[0].elapsed_time(s[1]) for s in spans) / len(spans)
    nb, fl = spans[0][2], spans[0][3]
    roof_ms = max(nb / HBM, fl / MFMA) * 1e3
    tot_b += nb; tot_roof += roof_ms
    layers.append({"layer": label, "ms": round(ms, 4), "MB": round(nb / 1e6, 1), "GB_per_s": round(nb / ms / 1e6, 1) if ms > 0 else None,
                   "frac_of_hbm_roof": round(nb / ms / 1e6 / 8000, 3) if ms > 0 else None,
                   "bound": "hbm" if nb / HBM >= fl / MFMA else "mfma"})
layers.sort(key=lambda r: -r["ms"])

# batched NMS at K >= 1000 (the other half of config 5)
def candidates(Bn, K, seed=2, size=480):
    g = torch.Generator().manual_seed(seed)
    c = torch.rand(Bn, K, 2, generator=g) * size
    wh = torch.exp(torch.rand(Bn, K, 2, generator=g) * (math.log(128.0) - math.log(8.0)) + math.log(8.0))
    return torch.cat([c - wh / 2, c + wh / 2], 2).round(), torch.rand(Bn, K, generator=g)

nms = {}
for K in (1024, 4096):
    boxes, scores = candidates(B, K)
    bd, sd = boxes.cuda(), scores.cuda()
    cnt = torch.full((B,), K, dtype=torch.int32, device="cuda")
    for _ in range(2): keep, kc = hp.nms_batched(bd, sd, cnt, 0.5)
    torch.cuda.synchronize()
    a.record()
    for _ in range(args.reps): keep, kc = hp.nms_batched(bd, sd, cnt, 0.5)
    e.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(e) / args.reps
    ref = O.nms(boxes[0], scores[0], 0.5)
    nms[f"K={K}"] = {"images": B, "ms": round(ms, 3), "images_per_s": round(B / ms * 1e3), "keep_set_of_image_0_equals_oracle": keep[0, : int(kc[0])].cpu().tolist() == ref.tolist()}

# CPU oracle on a bounded sample
nb = args.cpu_batch
torch.set_num_threads(min(16, os.cpu_count() or 1))
xc = x[:nb].cpu()
with torch.no_grad():
    MO.model_forward(P, xc[:1])
    t0 = time.perf_counter(); yc = MO.model_forward(P, xc); ct = time.perf_counter() - t0
err = (y[:nb].cpu() - yc).abs()
out = {"config": f"MobileNetV3-small backbone + 3x3 head, bf16 activations / fp32 accumulate, 3x{SIZE}x{SIZE}, bs {B}, 1 GPU, archive weights (g13)",
       "forward_ms": round(fwd_ms, 3), "imgs_per_s": round(B / fwd_ms * 1e3, 1),
       "algorithmic_MB_per_forward": round(tot_b / 1e6, 1), "hbm_roof_ms": round(tot_roof, 3), "frac_of_roof_whole_forward": round(tot_roof / fwd_ms, 3),
       "sum_of_layer_ms": round(sum(r["ms"] for r in layers), 3), "layers": layers, "batched_nms": nms,
       "max_abs_err_vs_oracle_first_images": float(err.max()), "parity": "unpinned (oracle = fp32 torch restatement; no reference output exists)",
       "cpu_oracle": {"imgs_per_s": round(nb / ct, 2), "what": f"forward of {nb} images, torch CPU fp32, {torch.get_num_threads()} threads"}}
s = json.dumps(out)
print(s)
if args.json:
    with open(args.json, "w") as f:
        f.write(s + "\n")
