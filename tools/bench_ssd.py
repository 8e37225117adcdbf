#!/usr/bin/env python3
"""SSD detection math at BASELINE.json config 4's batch (512 images, 4774 priors): target encode, ssd_loss
(hard negative mining + BCE + smooth-L1 + gradient) and ReduceSSDBoundingBoxes on the GPU, next to the
CPU oracle (torch ops, as the reference runs them) on a bounded sample.   python tools/bench_ssd.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fdet_amd
from fdet_amd import hotpath as hp
import oracle as O
from oracle import ssd_oracle as S

B, SIZE = 512, 480
P = hp.ssd_num_priors()
g = torch.Generator().manual_seed(0)
boxes = O.synthetic_boxes(B, SIZE, seed=2, max_faces=8)
pred = (torch.rand(B, P, 5, generator=g) * 0.98 + 0.01).cuda()
xdec = (torch.rand(B, P, 5, generator=g) * torch.tensor([0.51, 1, 1, 0.3, 0.3])).cuda()


def timeit(fn, reps):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


enc = hp.ssd_encode_targets(boxes, (SIZE, SIZE))
t_enc = timeit(lambda: hp.ssd_encode_targets(boxes, (SIZE, SIZE)), 5)
t_loss = timeit(lambda: hp.ssd_loss_fwd_bwd(pred, enc, 10), 10)
t_red = timeit(lambda: hp.ssd_reduce_bounding_boxes(xdec, 0.5, 0.5, SIZE, SIZE), 5)
# CPU oracle on a bounded sample (16 images)
torch.set_num_threads(min(16, os.cpu_count() or 1))
nb = 16
t0 = time.perf_counter(); [S.ssd_encode(b if b.numel() else torch.tensor([]), (SIZE, SIZE)) for b in boxes[:nb]]; c_enc = (time.perf_counter() - t0) / nb
pc, ec = pred[:nb].cpu(), enc[:nb].cpu()
t0 = time.perf_counter(); S.ssd_loss_and_grads(pc[:, :, 0], pc[:, :, 1:], ec[:, :, 0], ec[:, :, 1:], 10); c_loss = (time.perf_counter() - t0) / nb
xc = xdec[:nb].cpu()
t0 = time.perf_counter(); [S.reduce_ssd_bounding_boxes(xc[i], 0.5, 0.5, (3, SIZE, SIZE)) for i in range(nb)]; c_red = (time.perf_counter() - t0) / nb
loss_bytes = B * P * 5 * 4 * 3
print(json.dumps({
    "config": f"SSD detection math, {B} images x {P} priors (patch sizes 60/30/15/7), 3x480x480",
    "gpu_ms": {"encode_targets (incl. host box upload)": round(t_enc * 1e3, 3), "ssd_loss fwd+bwd": round(t_loss * 1e3, 3),
               "reduce_bounding_boxes (decode+NMS)": round(t_red * 1e3, 3)},
    "gpu_imgs_per_s": {"encode": round(B / t_enc), "loss": round(B / t_loss), "reduce": round(B / t_red)},
    "ssd_loss_algorithmic_mb": round(loss_bytes / 1e6, 1), "ssd_loss_gbs": round(loss_bytes / t_loss / 1e9, 1),
    "cpu_oracle_ms_per_image": {"encode": round(c_enc * 1e3, 3), "loss (batch of 16)": round(c_loss * 1e3, 3), "reduce": round(c_red * 1e3, 3)},
    "cpu_threads": torch.get_num_threads(), "kept_boxes_mean": float(hp.ssd_reduce_bounding_boxes(xdec, 0.5, 0.5, SIZE, SIZE)[1].float().mean())}))
