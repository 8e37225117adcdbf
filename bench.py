#!/usr/bin/env python3
"""Headline benchmark: training throughput of the YOLO face-detection hot path.

Workload (BASELINE.json metric): PoolResnet-medium (filters 64, 10 residual blocks, S=10) on
synthetic WIDER-Face-shaped batches 3x480x480, 256 images per GPU; one step = forward +
YoloLoss (batch sum) + backward + Adam, fp32.  Inputs and targets are resident in HBM before
the timed region.  Weak scaling: every rank trains on its own 256-image shard, gradients are
SUM-all-reduced over RCCL.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` without a launcher environment starts the N rank processes itself
(a `torch.distributed.run` child, started BEFORE this process touches the GPU) and exits with the
child's code.  Prints ONE JSON line on rank 0 (see the keys at the bottom).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

PEAK_FP32_MFMA_TFLOPS = 157.3        # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
SUSTAINED_BF16_MFMA_TFLOPS = 1780.0     # measured, see roofline.frac_of_sustained_bf16_mfma
PEAK_BF16_MFMA_TFLOPS = 2500.0       # dense bf16 MFMA peak
PEAK_HBM_GBS = 8000.0
# SURVEY.md 8d, PoolResnet F=64 @480^2 S=10, per image: compulsory activation traffic fwd+bwd and FLOPs
STEP_MB_PER_IMAGE_F64 = 26.90
STEP_GFLOP_PER_IMAGE_F64 = 3.0703
FWD_MB_PER_IMAGE_F64 = 11.81
FWD_GFLOP_PER_IMAGE_F64 = 1.0695
# timer groups that are launches of ONE kernel symbol (rocprofv3 --stats adds them up under that name)
SYMBOL_OF = {   # timer group -> kernel symbol (template instantiation) it launches on the default path; groups that share a symbol merge
    "chain_fwd@15x15": "k_block_chain_ps<false,false>@15x15", "chain_bwd@15x15": "k_block_chain_ps<true,false>@15x15",
    "conv3x3_wgrad@60x60": "k_wgrad3x3_ps<64,true,false>@60x60", "conv3x3_wgrad@30x30": "k_wgrad3x3_ps<32,true,false>@30x30",
    "conv3x3_wgrad@15x15": "k_wgrad3x3_ps<16,true,false>@15x15",
    "conv3x3_fwd@60x60": "k_conv3x3_ps<FWD_FULL,64,false>@60x60", "conv3x3_fwd@30x30": "k_conv3x3_ps<FWD_FULL,32,false>@30x30",
    "conv3x3_fwd_pool@60x60": "k_conv3x3_ps<FWD_POOL,64,false>@60x60", "conv3x3_fwd_pool@30x30": "k_conv3x3_ps<FWD_POOL,32,false>@30x30",
    "conv3x3_dgrad@60x60": "k_conv3x3_ps<DGRAD_ACT,64,false>@60x60", "conv3x3_dgrad@30x30": "k_conv3x3_ps<DGRAD_ACT,32,false>@30x30",
    "conv3x3_dgrad_unpool@60x60": "k_conv3x3_ps<DGRAD_ADDPOOL,64,false>@60x60", "conv3x3_dgrad_unpool@30x30": "k_conv3x3_ps<DGRAD_ADDPOOL,32,false>@30x30",
}


_T0 = time.perf_counter()


def _log(msg: str) -> None:
    """progress on stderr (the driver reads stdout's one JSON line; a long leg must not look hung)"""
    print(f"[bench +{time.perf_counter() - _T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


def self_launch(args) -> int:
    """Start N fresh rank processes (one per GPU) as a child `torch.distributed.run`; nothing in THIS
    process has initialised the GPU (device_count() does not on this image)."""
    ndev = torch.cuda.device_count()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if ndev < args.gpus and not env.get("FDET_SINGLE_DEVICE"):
        print(f"bench.py: --gpus {args.gpus} but only {ndev} GPU(s) are visible.  To rehearse the multi-rank path "
              "on one device set FDET_SINGLE_DEVICE=1 FDET_DIST_BACKEND=gloo.", file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def synth_batch(B, size, S, seed, device):
    """x = rand(B,3,480,480); targets encoded ON THE GPU from synthetic
    integer boxes, n ~ U{0,1,2} per image."""
    from fdet_amd import hotpath as hp
    from fdet_amd.datasets.synthetic import synthetic_boxes
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, 3, size, size, generator=g)
    boxes = synthetic_boxes(B, size, seed=seed + 1)
    return x.to(device), hp.encode_targets(boxes, (size, size), S, device=device), boxes


def cpu_model_name() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def effective_cpus() -> int:
    from fdet_amd.hostinfo import effective_cpus as f
    return f()


def cpu_baseline(filters, size, S, sample_bs, steps, frames=100, warm=20):
    """The CPU oracle (a port of the reference's CPU path on stock torch ops) timed on this box's host cores on
    a bounded sample of the same workload: (1) the headline training step, (2) BASELINE.json config 1, the
    reference's demo path (demo_model.py:17-21: uint8 frame stacked twice -> /255 -> conv stack -> decode -> NMS).
    BASELINE.md section 3 plans `torch.set_num_threads(os.cpu_count())`: `value` is the run on every core this process is
    GRANTED (effective_cpus(): affinity and cgroup quota; os.cpu_count() itself is reported as host_logical_cpus); the
    16-thread run of the earlier rounds is kept beside it (`threads16`).  A leg whose warm-up step alone exceeds its time
    budget is abandoned and reported as such instead of stalling the bench."""
    import oracle as O
    ncpu = effective_cpus()
    spec = O.poolresnet_spec(filters, (3, size, size), S)
    g = torch.Generator().manual_seed(0)
    x = torch.rand(sample_bs, 3, size, size, generator=g)
    y = torch.stack([O.encode_targets(b, (size, size), S) for b in O.synthetic_boxes(sample_bs, size, seed=1)])
    masks = O.make_dropout_masks(spec, sample_bs, seed=2)

    def train_leg(threads, n_steps, budget_s=15.0):
        """n_steps timed steps, cut short once `budget_s` seconds of CPU work are spent (at least one step)"""
        torch.set_num_threads(threads)
        P = O.init_params(spec, seed=0)
        state = {"exp_avg": {k: torch.zeros_like(v) for k, v in P.items()},
                 "exp_avg_sq": {k: torch.zeros_like(v) for k, v in P.items()}}
        t0 = time.perf_counter()
        O.train_step(spec, P, state, 1, x, y, masks)                   # warm-up
        if time.perf_counter() - t0 > budget_s:
            dtw = time.perf_counter() - t0
            _log(f"cpu_baseline: {threads} threads: the warm-up step took {dtw:.1f} s, leg abandoned")
            return sample_bs / dtw, dtw, 0
        t0 = time.perf_counter()
        done = 0
        for s in range(n_steps):
            O.train_step(spec, P, state, 2 + s, x, y, masks)
            done += 1
            if time.perf_counter() - t0 > budget_s:
                break
        dt = time.perf_counter() - t0
        _log(f"cpu_baseline: {threads} threads, {done} steps at batch {sample_bs} in {dt:.1f} s")
        return sample_bs * done / dt, dt, done

    v16, dt16, n16 = train_leg(min(16, ncpu), steps)
    if ncpu > 16:
        vall, dtall, nall = train_leg(ncpu, max(3, steps // 2))
    else:
        vall, dtall, nall = v16, dt16, n16
    # config 1: PoolResnet F=64 S=10 eval, thresholds 0.7 / 0.01 (the shipped archives' frozen values); a two-frame
    # forward is latency-bound on the host: timed at the faster of the two thread counts' natural choice (16)
    threads = min(16, ncpu)
    torch.set_num_threads(threads)
    P1 = O.init_params(spec, seed=0)
    u8 = torch.randint(0, 256, (3, size, size), dtype=torch.uint8, generator=torch.Generator().manual_seed(0))
    pair = torch.stack([u8, u8])
    with torch.no_grad():
        for _ in range(warm):
            O.predict_image0(spec, P1, pair, 0.7, 0.01)
        t1 = time.perf_counter()
        for _ in range(frames):
            O.predict_image0(spec, P1, pair, 0.7, 0.01)
        dt1 = (time.perf_counter() - t1) / frames
    return {"value": round(vall, 2), "unit": "imgs/s", "cores": ncpu, "kind": "port",
            "host_logical_cpus": os.cpu_count(), "threads_used": ncpu, "cpu_model": cpu_model_name(),
            "sample": f"{nall} training steps (fwd+loss+bwd+Adam) at batch {sample_bs}, "
                      f"oracle.train_step, torch CPU fp32, torch.set_num_threads({ncpu}), {dtall:.1f} s",
            "threads16": {"value": round(v16, 2), "unit": "imgs/s", "cores": min(16, ncpu),
                          "sample": f"{n16} training steps at batch {sample_bs}, {min(16, ncpu)} threads, {dt16:.1f} s"},
            "config1_demo_path": {"ms_per_frame": round(dt1 * 1e3, 3), "fps": round(1.0 / dt1, 1), "cores": threads,
                                  "sample": f"{frames} frames after {warm} warm-up, oracle.predict_image0 (2 stacked uint8 "
                                            "480x480 frames -> /255 -> PoolResnet-medium -> decode -> NMS of image 0, "
                                            "thresholds 0.7/0.01, random-init weights)"}}


def feed_legs(mm, model, B, size, S, device, steps):
    """SURVEY.md 8d/8f-1 under the driver's clock, same model and batch as the headline:
      step_with_metrics  fused_train_step(with_metrics=True): the decode + NMS of targets and predictions and the
                         IoU / recall / precision block every reference training_step runs (ModelMeta.py:170-218), inputs
                         resident in HBM;
      feed_inclusive     the uint8 batch (177 MB) crosses PCIe EVERY step from pinned host memory through U8BatchFeeder
                         (3 slots, copy stream, /255 on the device), overlapped with the previous step; no metrics;
      fit_path           trainer.fit() itself over pinned uint8 host batches: feed + step + metrics, the product training
                         loop (counterpart of Trainer.fit, train_model.py:47-60).
    None of these is `value`."""
    from fdet_amd.datasets.feed import U8BatchFeeder
    from fdet_amd.datasets.synthetic import synthetic_boxes
    from fdet_amd import hotpath as hp
    from fdet_amd import trainer
    out = {}
    x, y, _ = synth_batch(B, size, S, seed=100, device=device)
    _log("feed legs: step with metrics")
    dt, _ = _time_steps(lambda: mm.fused_train_step(x, y, with_metrics=True), 2, steps)
    out["step_with_metrics_ms"] = round(dt * 1e3, 3)
    dt0, _ = _time_steps(lambda: mm.fused_train_step(x, y), 2, steps)
    out["step_without_metrics_ms_same_loop"] = round(dt0 * 1e3, 3)
    del x
    g = torch.Generator().manual_seed(1)
    _log("feed legs: PCIe-inclusive step")
    feeder = U8BatchFeeder((B, 3, size, size), (size, size), device, target_shape=(B, 5, S, S), depth=3)
    y_host = y.cpu()
    frames = torch.randint(0, 256, (B, 3, size, size), dtype=torch.uint8, generator=g)
    for i in range(3):                                     # the three pinned slots are filled once (a loader decodes into them)
        pin, ypin = feeder.host_buffers()
        pin.copy_(frames.roll(i, 0)); ypin.copy_(y_host)
        feeder.submit()
        xf, yf, tok = feeder.get(); mm.fused_train_step(xf, yf); feeder.release(tok)

    def fed_step():
        feeder.submit()                                    # PCIe copy + /255 of the NEXT slot on the copy stream
        xf, yf, tok = feeder.get(); r = mm.fused_train_step(xf, yf); feeder.release(tok)
        return r
    dtf, _ = _time_steps(fed_step, 3, steps)
    out["feed_inclusive"] = {"ms_per_step": round(dtf * 1e3, 3), "imgs_per_s": round(B / dtf, 1),
                             "h2d_mb_per_step": round(B * 3 * size * size / 1e6, 1),
                             "what": "uint8 frames copied from pinned host memory every step + /255 on the device, copy overlapped "
                                     "with compute (U8BatchFeeder, 3 slots); fwd + YoloLoss + bwd + Adam"}
    del feeder
    # trainer.fit over host batches in pinned memory (what DataLoader(pin_memory=True) yields)
    nb = max(4, steps)
    _log("feed legs: trainer.fit")
    batches = []
    for i in range(2):
        batches.append((frames.roll(i + 1, 0).pin_memory(), y_host, None))
    seq = [batches[i % 2] for i in range(nb)]
    marks = []

    def on_step(i, train, o):
        if i == 1 or i == nb - 1:                          # steps 2 .. nb-1 timed (the first one builds the feeder)
            torch.cuda.synchronize(); marks.append(time.perf_counter())
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        old_log = mm.log_path
        mm.log_path = os.devnull
        trainer.fit(mm, seq, None, epochs=1, on_step=on_step)
        mm.log_path = old_log
    dtt = (marks[1] - marks[0]) / (nb - 2)
    out["fit_path"] = {"ms_per_step": round(dtt * 1e3, 3), "imgs_per_s": round(B / dtt, 1),
                       "what": "trainer.fit(): pinned uint8 host batches -> U8BatchFeeder -> fused_train_step(with_metrics=True)"}
    return out


def infer_bench(model, size, device, frames=100, warm=20):
    """BASELINE.json: "infer FPS incl. NMS", at the thresholds of config 1 (0.7 / 0.01, the shipped archives' frozen values,
    the same the host leg of cpu_baseline runs).  (1) the reference's demo path (demo_model.py:17-21): one uint8 frame
    stacked twice, forward(predict=1) = /255 -> conv stack -> sigmoid -> decode -> NMS of image 0, result read back by the
    host every frame; (2) batched serving: 256 uint8 frames per call, decode + NMS of every image on the device, counts
    read back once per batch."""
    model.eval()
    red = model.reduce_bounding_boxes
    old_thr = (red.probability_threshold, red.iou_threshold)
    red.probability_threshold, red.iou_threshold = 0.7, 0.01
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
            model.non_max_suppression(model.forward_frames(big))
        torch.cuda.synchronize()
        # median of 7 individually timed batches (two of five round-4 runs showed 17-20 ms batches in this leg right after the
        # preceding legs' buffers had gone back to the allocator, a third 1.72 ms with identical code: the median keeps one
        # allocator hiccup out of the figure, the spread is reported)
        per_batch = []
        for _ in range(7):
            t0 = time.perf_counter()
            outs = model.non_max_suppression(model.forward_frames(big))
            torch.cuda.synchronize()
            per_batch.append(time.perf_counter() - t0)
        per_batch.sort()
        dtb = per_batch[len(per_batch) // 2]
        # the same batch as ONE HIP graph (BaseModel.graphed_predict: /255, conv stack, decode + NMS captured once; a replay
        # is one launch) -- what a serving loop with a fixed batch shape would run
        graph_ms = None
        try:
            gp256 = model.graphed_predict(big)
            for _ in range(2):
                gp256(big)
            torch.cuda.synchronize()
            gb = []
            for _ in range(7):
                t0 = time.perf_counter()
                gp256(big)
                torch.cuda.synchronize()
                gb.append(time.perf_counter() - t0)
            gb.sort()
            graph_ms = gb[len(gb) // 2] * 1e3
            del gp256
        except Exception:                                    # noqa: BLE001
            graph_ms = None
        # the same batches in precision16 (engine.set_precision("bf16"): one MFMA pass; maps within ~1e-2 of the fp32-grade
        # ones, so boxes near a threshold may differ -- a second figure, opt-in for serving)
        p16_ms = None
        try:
            model.engine.set_precision("bf16")
            for _ in range(2):
                model.non_max_suppression(model.forward_frames(big))
            torch.cuda.synchronize()
            pb = []
            for _ in range(5):
                t0 = time.perf_counter()
                model.non_max_suppression(model.forward_frames(big))
                torch.cuda.synchronize()
                pb.append(time.perf_counter() - t0)
            pb.sort()
            p16_ms = pb[len(pb) // 2] * 1e3
        except Exception:                                    # noqa: BLE001 (no PS path for this model: no figure)
            p16_ms = None
        finally:
            if getattr(model.engine, "p16", False):
                model.engine.set_precision("bf16x3")
        # per-kernel table of one batched forward (HIP events on the launch stream)
        from fdet_amd.convstack import KernelTimer
        kt = KernelTimer(); model.engine.timer = kt
        model.forward_frames(big)
        model.engine.timer = None
        batched_kernels = {k: round(v[1], 4) for k, v in sorted(kt.summary().items(), key=lambda kv: -kv[1][1])}
    red.probability_threshold, red.iou_threshold = old_thr
    model.train()
    # roofline of the batched leg (per image): forward 1.0695 GFLOP (x3 bf16 passes), activations 11.81 MB,
    # plus the uint8 frame read (0.69 MB); the fp32 image (2 x 2.76 MB written and read back until round 4) is no longer part
    # of the path: forward_frames() hands the uint8 frames to the stem, which divides by 255 in its staging
    t_mfma = 256 * FWD_GFLOP_PER_IMAGE_F64 * 3.0 / (PEAK_BF16_MFMA_TFLOPS * 1e3) * 1e3
    fused_u8 = bool(model.engine.u8_frames_ok())
    mb = 256 * (FWD_MB_PER_IMAGE_F64 + 0.69 + (0.0 if fused_u8 else 2 * 2.76))
    t_hbm = mb / 1e3 / PEAK_HBM_GBS * 1e3
    return {"demo_path_ms_per_frame": round(dt1 * 1e3, 3), "demo_path_fps": round(1.0 / dt1, 1),
            "demo_path_hipgraph_ms_per_frame": round(dtg * 1e3, 3), "demo_path_hipgraph_fps": round(1.0 / dtg, 1),
            "batched_fps": round(256 / dtb, 1), "batched_ms_per_256": round(dtb * 1e3, 3), "batched_kernels_ms": batched_kernels,
            "batched_hipgraph": None if graph_ms is None else {"ms_per_256": round(graph_ms, 3), "fps": round(256 / (graph_ms * 1e-3), 1)},
            "batched_precision16": None if p16_ms is None else {"ms_per_256": round(p16_ms, 3), "fps": round(256 / (p16_ms * 1e-3), 1)},
            "batched_ms_min_max": [round(per_batch[0] * 1e3, 3), round(per_batch[-1] * 1e3, 3)],
            "batched_roofline": {"bound": "hbm" if t_hbm >= t_mfma else "mfma", "hbm_floor_ms": round(t_hbm, 3),
                                 "mfma_floor_ms": round(t_mfma, 3), "algorithmic_mb_per_256": round(mb, 1),
                                 "frac_of_max_floor": round(max(t_hbm, t_mfma) / (dtb * 1e3), 4)},
            "demo_path_note": "launch-bound (2 frames: ~20 launches, one count read-back): no roofline quoted",
            "u8_frames_fused_into_stem": fused_u8,
            "what": "uint8 3x480x480 frames -> /255 -> PoolResnet-medium -> decode -> greedy NMS (thresholds 0.7/0.01 = config 1, "
                    "random-init weights); demo path = 2 stacked frames per call, boxes of image 0 read by the host"}


def precision16_leg(B, size, S, F_, device, steps, warm):
    """SECOND key, never the headline: the same training step in `precision16` (engine.set_precision("bf16")): bf16
    activations and weights, ONE MFMA pass, fp32 accumulation / epilogues / master weights / optimizer -- the arithmetic of
    the reference's own Trainer(precision=16) (train_model.py:50).  This is the only regime in which the north star's
    "fraction of the HBM roofline" is defined for this model: 13.45 MB per image of 16-bit activation
    traffic against 0.31 ms of MFMA work.  Pinned by tests/test_gpu_p16.py against the reference run under
    torch.autocast("cpu", bfloat16) (fixture g17)."""
    from fdet_amd.models import ModelMeta
    from fdet_amd.models.PoolResnet import PoolResnet
    from fdet_amd.convstack import KernelTimer
    torch.manual_seed(0)
    model = PoolResnet(filters=F_, input_shape=(3, size, size), num_of_patches=S, num_of_residual_blocks=10).to(device).train()
    model.engine.set_precision("bf16")
    mm = ModelMeta(model=model, lr=1e-4)
    mm.configure_optimizers()
    x, y, _ = synth_batch(B, size, S, seed=100, device=device)
    dt, r = _time_steps(lambda: mm.fused_train_step(x, y), warm, steps)
    timer = KernelTimer()
    model.engine.timer = timer
    for _ in range(3):
        mm.fused_train_step(x, y)
    model.engine.timer = None
    per = timer.summary()
    ms = dt * 1e3
    gb16 = 0.5 * STEP_MB_PER_IMAGE_F64 * B / 1e3            # 13.45 MB per image in 16-bit activations
    gf = STEP_GFLOP_PER_IMAGE_F64 * B
    t_hbm, t_mfma = gb16 / PEAK_HBM_GBS * 1e3, gf / (PEAK_BF16_MFMA_TFLOPS * 1e3) * 1e3
    kern = {k: round(tot / 3, 4) for k, (n_l, tot, fl, nb) in sorted(per.items(), key=lambda kv: -kv[1][1])}
    return {"ms_per_step": round(ms, 3), "imgs_per_s": round(B / dt, 1), "dtype": "bf16 (one MFMA pass, fp32 accumulate, fp32 master weights)",
            "finite_loss": bool(torch.isfinite(r[0]).all()), "final_loss": round(float(r[0]), 4),
            "step_roofline": {"algorithmic_gb_per_step_16bit": round(gb16, 3), "hbm_floor_ms": round(t_hbm, 3),
                              "mfma_floor_ms": round(t_mfma, 3), "hbm_frac": round(gb16 / (ms * 1e-3) / PEAK_HBM_GBS, 4),
                              "frac_of_max_floor": round(max(t_hbm, t_mfma) / ms, 4),
                              "note": "the stem still reads the fp32 input (2.76 MB per image and pass) and the pooled-gradient "
                                      "routing still writes both planes: traffic above the 16-bit figure"},
            "kernels_ms_per_step": kern,
            "what": "same model, batch and step as the headline with engine.set_precision('bf16'); a second key, not `value`"}


def _time_steps(fn, warm, steps):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps, r


def extra_configs(device):
    """BASELINE.json configs 3, 4 and 5 on ONE GPU under the same clock as the headline (a few steps each): ms, img/s,
    algorithmic work and the fraction of its floor.  The multi-GPU legs of configs 3 / 4 are the
    driver's N > 1 runs of the headline path; these are the per-GPU shares."""
    import numpy as np
    import warnings
    from fdet_amd import hotpath as hp
    from fdet_amd.datasets.synthetic import synthetic_boxes
    from fdet_amd.models import ModelMeta
    from fdet_amd.models.ModelMetaSSD import ModelMetaSSD
    from fdet_amd.models.Resnet import Resnet
    from fdet_amd.models.SSD import SSD
    out = {}
    g = torch.Generator().manual_seed(3)

    def floors(gflop, mb, passes, ms):
        t_mfma = passes * gflop / (PEAK_BF16_MFMA_TFLOPS * 1e3) * 1e3
        t_hbm = (mb / 1e3 / PEAK_HBM_GBS * 1e3) if mb else 0.0
        return {"algorithmic_gflop": round(gflop, 1), "algorithmic_mb": round(mb, 1) if mb else None,
                "mfma_floor_ms": round(t_mfma, 3), "hbm_floor_ms": round(t_hbm, 3) if mb else None,
                "frac_of_max_floor": round(max(t_mfma, t_hbm) / ms, 4)}

    # config 2: the headline model at batch 64 on one GPU (BASELINE.json configs[1] as written: "bs=64 one training step")
    try:
        from fdet_amd.models.PoolResnet import PoolResnet
        from fdet_amd.convstack import KernelTimer
        B, size, S = 64, 480, 10
        torch.manual_seed(0)
        model = PoolResnet(filters=64, input_shape=(3, size, size), num_of_patches=S, num_of_residual_blocks=10).to(device).train()
        mm = ModelMeta(model=model, lr=1e-4); mm.configure_optimizers()
        x = torch.rand(B, 3, size, size, generator=g).to(device)
        y = hp.encode_targets(synthetic_boxes(B, size, seed=4), (size, size), S, device=device)
        dt, r = _time_steps(lambda: mm.fused_train_step(x, y), 3, 10)
        timer = KernelTimer(); model.engine.timer = timer
        mm.fused_train_step(x, y)
        model.engine.timer = None
        out["config2_bs64_1gpu"] = {
            "workload": "PoolResnet-medium 3x480x480, bs 64, fwd + YoloLoss + bwd + Adam (the headline kernels at a quarter of the batch)",
            "ms_per_step": round(dt * 1e3, 3), "imgs_per_s": round(B / dt, 1), "finite_loss": bool(torch.isfinite(r[0]).all()),
            **floors(B * STEP_GFLOP_PER_IMAGE_F64, B * STEP_MB_PER_IMAGE_F64 if "STEP_MB_PER_IMAGE_F64" in globals() else 0.0, 3.0, dt * 1e3),
            "kernels_ms_per_step": {k: round(tot, 4) for k, (n_l, tot, fl, nb) in sorted(timer.summary().items(), key=lambda kv: -kv[1][1])[:8]}}
        del model, mm, x, y
    except Exception as e:                                   # noqa: BLE001
        out["config2_bs64_1gpu"] = {"error": repr(e)[:300]}
    # config 3: Resnet-64 at 640^2, S=20, 32 images per GPU (global batch 256 on 8 GPUs)
    try:
        B, size, S = 32, 640, 20
        torch.manual_seed(0)
        model = Resnet(filters=64, input_shape=(3, size, size), num_of_patches=S, num_of_residual_blocks=10).to(device).train()
        mm = ModelMeta(model=model, lr=1e-4); mm.configure_optimizers()
        x = torch.rand(B, 3, size, size, generator=g).to(device)
        y = hp.encode_targets(synthetic_boxes(B, size, seed=4), (size, size), S, device=device)
        dt, r = _time_steps(lambda: mm.fused_train_step(x, y), 2, 5)
        out["config3_1gpu"] = {"workload": "Resnet-64 640^2 S=20, bs 32 per GPU, fwd + YoloLoss + bwd + Adam", "ms_per_step": round(dt * 1e3, 3),
                               "imgs_per_s": round(B / dt, 1), "finite_loss": bool(torch.isfinite(r[0]).all()),
                               **floors(B * 61.9384, B * 487.75, 3.0, dt * 1e3)}
        del model, mm, x, y
    except Exception as e:                                   # noqa: BLE001 (a leg must not take the headline line down)
        out["config3_1gpu"] = {"error": repr(e)[:300]}
    # PoolResnet "large" (filters 128): what the reference's train_model.py:17,27-31 actually trains.  Its layers run on the
    # round-2 fp32-I/O bf16x3 kernels (the PS machinery is 64-channel only): reported so that the gap is on record.
    try:
        from fdet_amd.models.PoolResnet import PoolResnet
        from fdet_amd.convstack import KernelTimer
        B, size, S = 256, 480, 10
        torch.manual_seed(0)
        model = PoolResnet(filters=128, input_shape=(3, size, size), num_of_patches=S, num_of_residual_blocks=10).to(device).train()
        mm = ModelMeta(model=model, lr=1e-4); mm.configure_optimizers()
        x = torch.rand(B, 3, size, size, generator=g).to(device)
        y = hp.encode_targets(synthetic_boxes(B, size, seed=4), (size, size), S, device=device)
        dt, r = _time_steps(lambda: mm.fused_train_step(x, y), 2, 5)
        timer = KernelTimer(); model.engine.timer = timer
        mm.fused_train_step(x, y)
        model.engine.timer = None
        per = timer.summary()
        out["poolresnet_large_F128_1gpu"] = {
            "workload": "PoolResnet-large (filters 128, 10 blocks, S=10) 3x480x480, bs 256, fwd + YoloLoss + bwd + Adam",
            "ms_per_step": round(dt * 1e3, 3), "imgs_per_s": round(B / dt, 1), "finite_loss": bool(torch.isfinite(r[0]).all()),
            "ps_path": bool(model.engine.ps), **floors(B * 11.7145, B * 48.27, 3.0, dt * 1e3),
            "kernels_ms_per_step": {k: round(tot, 4) for k, (n_l, tot, fl, nb) in sorted(per.items(), key=lambda kv: -kv[1][1])[:8]}}
        del model, mm, x, y
        torch.cuda.empty_cache()
    except Exception as e:                                   # noqa: BLE001
        out["poolresnet_large_F128_1gpu"] = {"error": repr(e)[:300]}
    # config 4: SSD filters 16, 4774 priors, hard-negative ratio 10, 64 images per GPU (global 512 on 8)
    try:
        B, size = 64, 480
        torch.manual_seed(0)
        model = SSD(filters=16, input_shape=(3, size, size)).to(device).train()
        mm = ModelMetaSSD(model=model, lr=1e-4); mm.configure_optimizers()
        x = torch.rand(B, 3, size, size, generator=g).to(device)
        y = hp.ssd_encode_targets(synthetic_boxes(B, size, seed=2), (size, size), device=device)
        dt, r = _time_steps(lambda: mm.fused_train_step(x, y), 2, 5)
        k4 = None
        try:                                                 # per-kernel table of one step (HIP events on the launch stream)
            from fdet_amd.convstack import KernelTimer
            eng4 = model.engine
            timer4 = KernelTimer(); eng4.timer = timer4
            mm.fused_train_step(x, y)
            eng4.timer = None
            k4 = {k: round(tot, 4) for k, (n_l, tot, fl, nb) in sorted(timer4.summary().items(), key=lambda kv: -kv[1][1])[:10]}
        except Exception:                                    # noqa: BLE001 (the SSD engine may not carry a timer)
            k4 = None
        # forward 7.18 GMAC per image; backward = data + weight gradients of the same layers
        out["config4_1gpu"] = {"workload": "SSD filters 16, 3x480x480, 4774 priors, bs 64 per GPU, fwd + ssd_loss + bwd + Adam",
                               "ms_per_step": round(dt * 1e3, 3), "imgs_per_s": round(B / dt, 1),
                               "finite_loss": bool(torch.isfinite(r[0]).all()), **floors(B * 3 * 14.36, 0.0, 3.0, dt * 1e3),
                               "kernels_ms_per_step": k4}
        del model, mm, x, y
    except Exception as e:                                   # noqa: BLE001
        out["config4_1gpu"] = {"error": repr(e)[:300]}
    # config 5: MobileNetV3-small backbone, bf16, bs 256 forward + batched NMS over K = 1024 candidates per image
    try:
        from fdet_amd.models.MobilenetV3Backbone import MobilenetV3Backbone
        z = np.load(os.path.join(ROOT, "tests", "golden", "g13_mobilenet_weights.npz"))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            net = MobilenetV3Backbone(64, (3, 480, 480), 15, pretrained=False)
        net.load_state_dict({k: torch.from_numpy(z[k]) for k in z.files})
        net = net.to(device).eval()
        B, K = 256, 1024
        x = torch.rand(B, 3, 480, 480, generator=g).to(device)
        with torch.no_grad():
            dt, _ = _time_steps(lambda: net(x), 2, 5)
        eng = net._packed_engine()
        rows = {}

        class _T:                                             # algorithmic bytes per layer as the engine declares them
            def __call__(self, label, nbytes, flops):
                rows[label] = (nbytes, flops)
                return self
            def __enter__(self): return self
            def __exit__(self, *e): return False
        eng.timer = _T()
        with torch.no_grad():
            net(x)
        eng.timer = None
        mb = sum(v[0] for v in rows.values()) / 1e6
        gf = sum(v[1] for v in rows.values()) / 1e9
        c = torch.rand(B, K, 2, generator=g) * 480
        import math
        wh = torch.exp(torch.rand(B, K, 2, generator=g) * (math.log(128.0) - math.log(8.0)) + math.log(8.0))
        bd = torch.cat([c - wh / 2, c + wh / 2], 2).round().to(device)
        sd = torch.rand(B, K, generator=g).to(device)
        cnt = torch.full((B,), K, dtype=torch.int32, device=device)
        dtn, _ = _time_steps(lambda: hp.nms_batched(bd, sd, cnt, 0.5), 2, 5)
        # the same backbone's TRAINING step (round 4: fp32, BatchNorm batch statistics, correctness-first kernels): reported so
        # that the gap to the inference engine is on record
        train_ms = None
        try:
            from fdet_amd.models import ModelMeta as _MM
            Bt = 32
            net.train()
            mmt = _MM(model=net, lr=1e-4)
            (opt_t,), _ = mmt.configure_optimizers()
            xt = torch.rand(Bt, 3, 480, 480, generator=g).to(device)
            bx = synthetic_boxes(Bt, 480, seed=9)
            yt = hp.encode_targets(bx, (480, 480), 15, device=device)

            def tstep():
                o = mmt.training_step((xt, yt, bx), 0)
                opt_t.zero_grad(); o["loss"].backward(); opt_t.step()
                return o["loss"]
            dtt, lt = _time_steps(tstep, 1, 3)
            train_ms = {"ms_per_step_bs32": round(dtt * 1e3, 2), "imgs_per_s": round(Bt / dtt, 1), "finite_loss": bool(torch.isfinite(lt)),
                        "what": "ModelMeta.training_step (metrics block on) + loss.backward() + SAMSGD.step(): the reference's Lightning-style path"}
            # the same update without the autograd graph / per-parameter gradient copies / metrics (ModelMeta.fused_train_step)
            dtf, lf = _time_steps(lambda: mmt.fused_train_step(xt, yt)[0], 2, 5)
            train_ms["fused"] = {"ms_per_step_bs32": round(dtf * 1e3, 2), "imgs_per_s": round(Bt / dtf, 1), "finite_loss": bool(torch.isfinite(lf).all())}
            net.eval()
        except Exception as e:                               # noqa: BLE001
            train_ms = {"error": repr(e)[:200]}
        out["config5_1gpu"] = {"workload": "MobileNetV3-small backbone + head, bf16 activations, bs 256 forward; batched NMS K=1024 per image",
                               "training_step": train_ms,
                               "forward_ms": round(dt * 1e3, 3), "imgs_per_s": round(B / dt, 1), "nms_ms_per_256x1024": round(dtn * 1e3, 3),
                               **floors(gf, mb, 1.0, dt * 1e3)}
    except Exception as e:                                   # noqa: BLE001
        out["config5_1gpu"] = {"error": repr(e)[:300]}
    return out


def kernel_source_hash() -> str:
    """sha256 over the HIP sources of the kernels the headline step runs (conv stack, stem, head, chain, tails, loss,
    optimiser; not the SSD / pointwise / MobileNet / preprocessing units): the PMC traffic file below was measured on
    ONE build and is only quoted for that build."""
    d = os.path.join(ROOT, "pytorch-face-detection-from-scratch_amd", "csrc")
    other = ("fdet_mobilenet", "fdet_pointwise", "fdet_preproc")
    h = hashlib.sha256()
    for fn in sorted(os.listdir(d)):
        if (fn.endswith((".hip", ".h", ".inc")) or fn == "Makefile") and not fn.startswith(other):
            h.update(fn.encode())
            h.update(open(os.path.join(d, fn), "rb").read())
    return h.hexdigest()


def pmc_traffic_for(kernel_group: str, B: int, F_: int):
    """HBM bytes per launch of a kernel group from the committed rocprofv3 PMC passes (profiles/*pmc_traffic*.json
    written by tools/pmc_traffic.py with the source hash of the build it profiled) -- null when the sources have
    changed since, or for another batch / width."""
    path = os.path.join(ROOT, "profiles", "r04_pmc_traffic.json")
    try:
        tj = json.load(open(path))
        if tj.get("csrc_sha256") != kernel_source_hash() or tj.get("batch") != B or tj.get("filters") != F_:
            return None
        return tj["kernels"][kernel_group]["hbm_bytes_per_launch"]
    except Exception:                                        # noqa: BLE001 (missing file / key: no traffic figure)
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU")
    ap.add_argument("--filters", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-inference", action="store_true", help="skip the inference leg (profiling runs: its launches share kernel symbols with training)")
    ap.add_argument("--no-configs", action="store_true", help="skip the config 3 / 4 / 5 legs")
    ap.add_argument("--no-feed", action="store_true", help="skip the metrics-on / PCIe-inclusive / trainer.fit legs")
    ap.add_argument("--no-p16", action="store_true", help="skip the precision16 leg (second key)")
    ap.add_argument("--concat-ranks", type=int, default=0,
                    help="(tests) one process whose batch is the concatenation of the shards R ranks would draw")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))                        # N fresh rank processes; this one never touches the GPU

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
            print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; reporting n_gpus={world}",
                  file=sys.stderr)
        args.gpus = world
    device = torch.device("cuda", torch.cuda.current_device())

    import fdet_amd
    from fdet_amd.models import ModelMeta
    from fdet_amd.models.PoolResnet import PoolResnet
    from fdet_amd.convstack import KernelTimer
    from fdet_amd.hostinfo import limit_host_threads
    limit_host_threads()                                   # the box grants 16 of its 256 logical CPUs (hostinfo.py)

    size, S, B, F_ = 480, 10, args.batch, args.filters
    torch.manual_seed(0)                                   # train_model.py:13 (ranks are synchronised to rank 0's weights anyway)
    model = PoolResnet(filters=F_, input_shape=(3, size, size), num_of_patches=S, num_of_residual_blocks=10).to(device)
    model.train()
    mm = ModelMeta(model=model, lr=1e-4)
    mm.configure_optimizers()
    if args.concat_ranks > 1:                               # the batch an R-rank job trains on, in one process
        parts = [synth_batch(B // args.concat_ranks, size, S, seed=100 + r, device=device) for r in range(args.concat_ranks)]
        x, y = torch.cat([p_[0] for p_ in parts]), torch.cat([p_[1] for p_ in parts])
    else:
        x, y, _ = synth_batch(B, size, S, seed=100 + rank, device=device)

    for _ in range(args.warmup):
        mm.fused_train_step(x, y)
    if mm._reducer is not None and mm._reducer.enabled:
        mm._reducer.timing = []                              # HIP events around red.wait(): the exposed all-reduce time
    timer = KernelTimer()
    model.engine.timer = None if os.environ.get("FDET_BENCH_NO_TIMER") else timer
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # live per-kernel timing (HIP events on the launch stream) on the FIRST n_timed of the K timed steps: two event
    # records per launch cost ~0.13 ms per step (measured: 3.26 vs 3.12 ms with / without them), so the sample is
    # kept short; every timed step runs the identical work
    n_timed = min(args.steps, 3)
    for s_ in range(args.steps):
        if s_ == n_timed:
            model.engine.timer = None
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
    if world > 1:                                           # whole-job loss of the last step: the per-rank batch sums add up
        lt = lsum.detach().double().reshape(1).clone()
        dist.all_reduce(lt)
        loss_val = float(lt)
    else:
        loss_val = float(lsum)

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * B * args.steps / dt
        # ---- roofline of the dominant kernel group (live HIP-event timing over the timed region, on the stream the
        # kernels are launched on).  Each (kind, resolution) group is ONE kernel symbol in this model, so avg_launch_ms
        # is the average `rocprofv3 --stats` reports for that symbol when the same command is profiled with
        # --no-inference (the inference leg reuses the forward symbols).  Per group
        #   t_roof = max(algorithmic bytes / 8 TB/s, MFMA flops / peak)
        # where the bf16x3 arithmetic issues THREE bf16 MFMA passes per fp32 MAC (peak 2.5 PFLOP/s dense) and the exact
        # fp32 path one f32 MFMA pass (157.3 TFLOP/s); `bound` names the larger term.
        per = timer.summary()                                # name -> (launches, total ms, flops/launch, bytes/launch)
        if not per:                                          # FDET_BENCH_NO_TIMER=1 (development: the cost of the live timing itself)
            print(json.dumps({"ms_per_step_without_kernel_timer": round(ms, 3), "value": round(value, 1)}))
            return
        x3 = bool(model.engine.x3)
        mf_mult, mf_peak = (3.0, PEAK_BF16_MFMA_TFLOPS) if x3 else (1.0, PEAK_FP32_MFMA_TFLOPS)
        table = {}
        for k, (n_l, tot, fl, nb) in per.items():
            avg = tot / n_l
            t_hbm = nb / (PEAK_HBM_GBS * 1e9) * 1e3          # ms
            t_mfma = mf_mult * fl / (mf_peak * 1e12) * 1e3
            bound = "mfma" if t_mfma > t_hbm else "hbm"
            troof = max(t_hbm, t_mfma)
            table[k] = {"ms_per_step": round(tot / n_timed, 4), "avg_launch_ms": round(avg, 4), "bound": bound,
                        "t_roof_ms": round(troof, 4), "frac": round(troof / avg, 4) if troof > 0 else None}
        # dominant kernel SYMBOL: groups that launch the same kernel are added up (what rocprofv3 --stats shows)
        symmap = SYMBOL_OF if getattr(model.engine, "ps", False) else {"chain_fwd@15x15": "k_block_chain_x3<false>@15x15", "chain_bwd@15x15": "k_block_chain_x3<false>@15x15"}
        sym = {}
        for k, (n_l, tot, fl, nb) in per.items():
            e = sym.setdefault(symmap.get(k, k), [0, 0.0, 0.0, 0.0, []])
            e[0] += n_l; e[1] += tot; e[2] += fl * n_l; e[3] += nb * n_l; e[4].append(k)
        dom = max(sym, key=lambda k: sym[k][1])
        n_l, tot_ms, fl_tot, nb_tot, dom_groups = sym[dom]
        fl, nb = fl_tot / n_l, nb_tot / n_l                   # per launch, averaged over the symbol's launches
        avg_ms = tot_ms / n_l
        t_hbm_d = nb / (PEAK_HBM_GBS * 1e9) * 1e3
        t_mfma_d = mf_mult * fl / (mf_peak * 1e12) * 1e3
        d = {"bound": "mfma" if t_mfma_d > t_hbm_d else "hbm"}
        if d["bound"] == "hbm":
            achieved, peak, unit = nb / (avg_ms * 1e-3) / 1e9, PEAK_HBM_GBS, "GB/s"
        else:
            achieved, peak, unit = mf_mult * fl / (avg_ms * 1e-3) / 1e12, mf_peak, "TFLOP/s"
        sum_roof = sum(v["t_roof_ms"] * (per[k][0] / n_timed) for k, v in table.items())
        sum_t = sum(v["ms_per_step"] for v in table.values())
        roof = {"bound": d["bound"], "kernel": dom, "groups": dom_groups, "achieved": round(achieved, 2), "peak": peak, "unit": unit,
                "frac": round(achieved / peak, 4), "traffic": pmc_traffic_for("+".join(dom_groups), B, F_),
                "avg_launch_ms": round(avg_ms, 4), "launches_per_step": n_l // n_timed, "timed_on_steps": n_timed,
                "algorithmic_mb_per_launch": round(nb / 1e6, 1), "algorithmic_gflop_per_launch": round(fl / 1e9, 3),
                "mfma_passes_per_mac": mf_mult,
                # every fp32 multiply-add counted ONCE (the three bf16 passes of the split are the price of fp32-grade
                # parity, not useful work): the same launch against the same peak
                "frac_single_count": round(fl / (avg_ms * 1e-3) / 1e12 / mf_peak, 4) if d["bound"] == "mfma" else round(achieved / peak, 4),
                # the rate this chip SUSTAINS on random bf16 operands under its power limit (tools/probe/mfma_shapes.py,
                # profiles/r02_c_mfma_shapes.json: 1.78 PFLOP/s of 32x32x16 MFMAs at ~2.0 GHz) -- context, not the roof
                "frac_of_sustained_bf16_mfma": round(mf_mult * fl / (avg_ms * 1e-3) / 1e12 / SUSTAINED_BF16_MFMA_TFLOPS, 4) if x3 else None,
                "hbm_frac": round(nb / (avg_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4) if nb else None}
        out = {
            "metric": "train imgs/sec (PoolResnet 480^2, bs=256 per GPU)", "value": round(value, 1), "unit": "imgs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (3x3 convs: bf16x3 split MFMA, fp32 accumulate)" if x3 else "f32", "data": "synthetic",
            "config": {"workload": f"PoolResnet-medium (filters {F_}, 10 blocks, S=10) 3x{size}x{size}, one training "
                                   "step = fwd + YoloLoss + bwd + Adam", "global_batch": world * B, "per_gpu_batch": B,
                       "parallelism": f"dp{world}"},
            "roofline": roof, "kernels": dict(sorted(table.items(), key=lambda kv: -kv[1]["ms_per_step"])),
            "final_loss": round(loss_val, 4),
            "sum_t_roof_over_sum_t": round(sum_roof / sum_t, 4) if sum_t > 0 else None,
            "n_ranks_seen": dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1,
        }
        if mm._reducer is not None and mm._reducer.timing:
            ev = mm._reducer.timing
            out["allreduce_ms_exposed"] = round(sum(a.elapsed_time(b) for a, b in ev) / len(ev), 4)
            out["allreduce_backend"] = dist.get_backend() if dist.is_initialized() else None
        if F_ == 64:
            gb = STEP_MB_PER_IMAGE_F64 * B / 1e3
            gf = STEP_GFLOP_PER_IMAGE_F64 * B
            out["step_roofline"] = {
                "algorithmic_gb_per_step": round(gb, 3), "achieved_gbs": round(gb / (ms * 1e-3), 1),
                "hbm_frac": round(gb / (ms * 1e-3) / PEAK_HBM_GBS, 4),
                "mfma_floor_ms": round(mf_mult * gf / (mf_peak * 1e3) * 1e3, 3), "hbm_floor_ms": round(gb / PEAK_HBM_GBS * 1e3, 3),
                "frac_of_max_floor": round(max(mf_mult * gf / (mf_peak * 1e3) * 1e3, gb / PEAK_HBM_GBS * 1e3) / ms, 4),
                "frac_of_max_floor_single_count": round(max(gf / (mf_peak * 1e3) * 1e3, gb / PEAK_HBM_GBS * 1e3) / ms, 4)}
        if world == 1 and not args.no_feed:
            out.update(feed_legs(mm, model, B, size, S, device, steps=min(args.steps, 10)))
        _log(f"headline timed: {ms:.3f} ms/step")
        if world == 1 and not args.no_p16 and F_ == 64:
            _log("precision16 leg")
            try:
                out["precision16"] = precision16_leg(B, size, S, F_, device, steps=args.steps, warm=args.warmup)
            except Exception as e:                           # noqa: BLE001 (a leg must not take the headline line down)
                out["precision16"] = {"error": repr(e)[:300]}
        if world == 1 and not args.no_inference:
            _log("inference leg")
            out["inference"] = infer_bench(model, size, device)
        if world == 1 and not args.no_configs:
            del x, y
            torch.cuda.empty_cache()
            _log("configs 3 / 4 / 5")
            out["configs"] = extra_configs(device)
        if world == 1 and not args.no_cpu_baseline:
            _log("cpu baseline")
            out["cpu_baseline"] = cpu_baseline(F_, size, S, sample_bs=64, steps=20)
        _log("done")
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
