#!/usr/bin/env python3
"""Headline benchmark: training throughput of the YOLO face-detection hot path.

Workload (BASELINE.json metric): PoolResnet-medium (filters 64, 10 residual blocks, S=10) on
synthetic WIDER-Face-shaped batches 3x480x480, 256 images per GPU; one step = forward +
YoloLoss (batch sum) + backward + Adam, fp32.  Inputs and targets are resident in HBM before
the timed region.  Weak scaling: every rank trains on its own 256-image shard, gradients are
SUM-all-reduced over RCCL.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (see the keys at the bottom).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

PEAK_FP32_MFMA_TFLOPS = 157.3        # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_HBM_GBS = 8000.0


def synth_batch(B, size, S, seed, device):
    """x = rand(B,3,480,480); targets encoded ON THE GPU from synthetic
    integer boxes, n ~ U{0,1,2} per image."""
    import oracle as O            # box generator only (host-side data synthesis)
    from fdet_amd import hotpath as hp
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, 3, size, size, generator=g)
    boxes = O.synthetic_boxes(B, size, seed=seed + 1)
    return x.to(device), hp.encode_targets(boxes, (size, size), S, device=device), boxes


def cpu_baseline(filters, size, S, sample_bs, steps):
    """The CPU oracle (a port of the reference's CPU path on stock torch ops) timed on this
    box's host cores on a bounded sample of the same workload."""
    import oracle as O
    threads = min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    spec = O.poolresnet_spec(filters, (3, size, size), S)
    P = O.init_params(spec, seed=0)
    state = {"exp_avg": {k: torch.zeros_like(v) for k, v in P.items()},
             "exp_avg_sq": {k: torch.zeros_like(v) for k, v in P.items()}}
    g = torch.Generator().manual_seed(0)
    x = torch.rand(sample_bs, 3, size, size, generator=g)
    y = torch.stack([O.encode_targets(b, (size, size), S) for b in O.synthetic_boxes(sample_bs, size, seed=1)])
    masks = O.make_dropout_masks(spec, sample_bs, seed=2)
    O.train_step(spec, P, state, 1, x, y, masks)                       # warm-up
    t0 = time.perf_counter()
    for s in range(steps):
        O.train_step(spec, P, state, 2 + s, x, y, masks)
    dt = time.perf_counter() - t0
    return {"value": round(sample_bs * steps / dt, 2), "unit": "imgs/s", "cores": threads, "kind": "port",
            "sample": f"{steps} training steps (fwd+loss+bwd+Adam) at batch {sample_bs}, oracle.train_step, "
                      f"torch CPU fp32, {threads} threads, {dt:.1f} s"}


def infer_bench(model, size, device, frames=100, warm=20):
    """BASELINE.json: "infer FPS incl. NMS".  (1) the reference's demo path (demo_model.py:17-21): one
    uint8 frame stacked twice, forward(predict=1) = /255 -> conv stack -> sigmoid -> decode -> NMS of
    image 0, result read back by the host every frame; (2) batched serving: 256 uint8 frames per
    call, decode + NMS of every image on the device, counts read back once per batch."""
    model.eval()
    g = torch.Generator().manual_seed(0)
    u8 = torch.randint(0, 256, (3, size, size), dtype=torch.uint8, generator=g)
    pair = torch.stack([u8, u8]).to(device)
    with torch.no_grad():
        for _ in range(warm):
            model(pair, predict=torch.tensor(1))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(frames):
            det = model(pair, predict=torch.tensor(1))
        torch.cuda.synchronize()
        dt1 = (time.perf_counter() - t0) / frames
        # the same path replayed from a HIP graph (one graph launch per frame instead of ~20 kernel launches)
        gp = model.graphed_predict(pair)
        for _ in range(warm):
            gp.first(pair)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(frames):
            det = gp.first(pair)
        torch.cuda.synchronize()
        dtg = (time.perf_counter() - t0) / frames
        big = torch.randint(0, 256, (256, 3, size, size), dtype=torch.uint8, generator=g).to(device)
        for _ in range(2):
            model.non_max_suppression(model(model._preprocess(big)))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            outs = model.non_max_suppression(model(model._preprocess(big)))
        torch.cuda.synchronize()
        dtb = (time.perf_counter() - t0) / reps
    model.train()
    return {"demo_path_ms_per_frame": round(dt1 * 1e3, 3), "demo_path_fps": round(1.0 / dt1, 1),
            "demo_path_hipgraph_ms_per_frame": round(dtg * 1e3, 3), "demo_path_hipgraph_fps": round(1.0 / dtg, 1),
            "batched_fps": round(256 / dtb, 1), "batched_ms_per_256": round(dtb * 1e3, 3),
            "what": "uint8 3x480x480 frames -> /255 -> PoolResnet-medium -> decode -> greedy NMS (thresholds 0.5/0.5, "
                    "random-init weights); demo path = 2 stacked frames per call, boxes of image 0 read by the host"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU")
    ap.add_argument("--filters", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-inference", action="store_true", help="skip the inference leg (profiling runs: its launches share kernel symbols with training)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # rehearsal on a one-GPU box: FDET_DIST_BACKEND=gloo FDET_SINGLE_DEVICE=1 (all ranks on cuda:0)
        backend = os.environ.get("FDET_DIST_BACKEND", "nccl")
        if os.environ.get("FDET_SINGLE_DEVICE"):
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(0)
    if args.gpus != world:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        args.gpus = world
    device = torch.device("cuda", torch.cuda.current_device())

    import fdet_amd
    from fdet_amd.models import ModelMeta
    from fdet_amd.models.PoolResnet import PoolResnet
    from fdet_amd.convstack import KernelTimer

    size, S, B, F_ = 480, 10, args.batch, args.filters
    torch.manual_seed(0)                                   # train_model.py:13; same init on every rank
    model = PoolResnet(filters=F_, input_shape=(3, size, size), num_of_patches=S, num_of_residual_blocks=10).to(device)
    model.train()
    mm = ModelMeta(model=model, lr=1e-4)
    mm.configure_optimizers()
    x, y, _ = synth_batch(B, size, S, seed=100 + rank, device=device)

    for _ in range(args.warmup):
        mm.fused_train_step(x, y)
    timer = KernelTimer()
    model.engine.timer = timer
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        lsum, _, _ = mm.fused_train_step(x, y)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    model.engine.timer = None
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    loss_val = float(lsum)

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * B * args.steps / dt
        # ---- roofline of the dominant kernel (live HIP-event timing over the timed region, on the
        # stream the kernels are launched on).  Each (kind, resolution) group is one kernel symbol in
        # this model (e.g. conv3x3_wgrad@60x60 = k_wgrad3x3_x3<2,4,false,false>, one launch per step
        # for both 60x60 layers; conv3x3_fwd@60x60 = k_conv3x3_x3_sb<2,2,4,FWD_FULL>), so
        # avg_launch_ms is the average `rocprofv3 --stats` reports for that symbol when the same
        # command is profiled with --no-inference (the inference leg reuses the forward symbols).
        # bf16x3 convs are HBM-bound (peak 8 TB/s); the exact-fp32 path is bound by the fp32 MFMA
        # rate (157.3 TFLOP/s).
        per = timer.summary()                                # name -> (launches, total ms, flops/launch, bytes/launch)
        x3 = bool(model.engine.x3)
        groups = {}
        for k, (n_l, tot, fl, nb) in per.items():
            kind, shape = k.split("@")
            g = k                                            # one kernel symbol per (kind, resolution) in this model
            a = groups.setdefault(g, [0, 0.0, 0.0, 0.0])
            a[0] += n_l; a[1] += tot; a[2] += fl * n_l; a[3] += nb * n_l
        dom = max(groups, key=lambda k: groups[k][1])
        n_l, tot_ms, fl_sum, nb_sum = groups[dom]
        avg_ms = tot_ms / n_l
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            if B == 256 and F_ == 64 and x3 and dom in tj:
                traffic = tj[dom]["hbm_bytes_per_launch"]
        except Exception:
            pass
        if x3 and nb_sum > 0:
            achieved = nb_sum / n_l / (avg_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(achieved / PEAK_HBM_GBS, 4), "traffic": traffic, "avg_launch_ms": round(avg_ms, 4),
                    "launches_per_step": n_l // args.steps, "algorithmic_mb_per_launch": round(nb_sum / n_l / 1e6, 1),
                    "algorithmic_gflop_per_launch": round(fl_sum / n_l / 1e9, 3)}
        else:
            achieved = fl_sum / n_l / (avg_ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "kernel": dom, "achieved": round(achieved, 2), "peak": PEAK_FP32_MFMA_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": traffic,
                    "avg_launch_ms": round(avg_ms, 4), "launches_per_step": n_l // args.steps,
                    "algorithmic_gflop_per_launch": round(fl_sum / n_l / 1e9, 3)}
        breakdown = {k: round(v[1] / args.steps, 3) for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])}
        out = {
            "metric": "train imgs/sec (PoolResnet 480^2, bs=256 per GPU)", "value": round(value, 1), "unit": "imgs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (3x3 convs: bf16x3 split MFMA, fp32 accumulate)" if x3 else "f32", "data": "synthetic",
            "config": {"workload": f"PoolResnet-medium (filters {F_}, 10 blocks, S=10) 3x{size}x{size}, one training "
                                   "step = fwd + YoloLoss + bwd + Adam", "global_batch": world * B, "per_gpu_batch": B,
                       "parallelism": f"dp{world}"},
            "roofline": roof, "kernel_ms_per_step": breakdown, "final_loss": round(loss_val, 4),
        }
        if world == 1 and not args.no_inference:
            out["inference"] = infer_bench(model, size, device)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(F_, size, S, sample_bs=64, steps=24)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
