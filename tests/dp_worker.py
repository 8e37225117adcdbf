"""One data-parallel rank of tests/test_gpu_dataparallel.py (started as a fresh child process; all ranks share
cuda:0, gloo carries the collectives): runs ModelMeta.fused_train_step on ITS shard of a global batch and, on
rank 0, saves the parameters and the all-reduced flat gradient."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import torch
import torch.distributed as dist


def build_case(F_, B, steps_seed=0):
    import oracle as O
    S, size = 10, 480
    spec = O.poolresnet_spec(F_, (3, size, size), S)
    x = torch.rand(B, 3, size, size, generator=torch.Generator().manual_seed(10))
    y = torch.stack([O.encode_targets(b, (size, size), S) for b in O.synthetic_boxes(B, size, seed=11)])
    masks = O.make_dropout_masks(spec, B, seed=12)
    return spec, x, y, masks


def run_rank(rank, world, F_, b_local, steps, live_dropout, param_seed):
    import fdet_amd  # noqa: F401
    import oracle as O
    from fdet_amd.models import ModelMeta
    from fdet_amd.models.PoolResnet import PoolResnet
    spec, x, y, masks = build_case(F_, world * b_local)
    P = O.init_params(spec, seed=param_seed)
    model = PoolResnet(F_, (3, 480, 480), 10)
    model.load_state_dict({k: v.clone() for k, v in P.items()})
    model = model.cuda().train()
    mm = ModelMeta(model=model, lr=1e-4)
    mm.configure_optimizers()
    lo, hi = rank * b_local, (rank + 1) * b_local
    if not live_dropout:
        model.set_dropout_masks({k: v[lo:hi] for k, v in masks.items()})
    xs, ys = x[lo:hi].cuda(), y[lo:hi].cuda()
    losses = []
    for s_ in range(steps):
        lsum, _, _ = mm.fused_train_step(xs, ys)
        losses.append(float(lsum))
        if s_ == 0:
            # the step-1 gradient: every rank started from rank 0's parameters and the same masks, so the all-reduced
            # gradient may differ from the single-process one by summation order only
            run_rank.grad_step1 = mm.opt._space().grad.cpu().clone()
    sp = mm.opt._space()
    return ({k: v.detach().cpu().clone() for k, v in model.state_dict().items()}, sp.grad.cpu().clone(), losses,
            mm._reducer.enabled)


def run_rank_ssd(rank, world, b_local, steps, live_dropout, param_seed):
    """SSD (filters 16) through ModelMetaSSD.fused_train_step: the batch-wide positive count is exchanged before the
    backward pass, the flat gradient all-reduced after it."""
    import fdet_amd  # noqa: F401
    import oracle as O
    from fdet_amd import hotpath as hp
    from fdet_amd.models.SSD import SSD
    from fdet_amd.models.ModelMetaSSD import ModelMetaSSD
    from oracle import ssd_model_oracle as SM
    fil, size, B = 16, 480, world * b_local
    P = SM.init_params(fil, seed=40 + param_seed)
    model = SSD(filters=fil, input_shape=(3, size, size))
    model.load_state_dict({k: v.clone() for k, v in P.items()})
    model = model.cuda().train()
    mm = ModelMetaSSD(model=model, lr=1e-4)
    mm.configure_optimizers()
    x = torch.rand(B, 3, size, size, generator=torch.Generator().manual_seed(41))
    boxes = O.synthetic_boxes(B, size, seed=42, max_faces=5)
    boxes[B - 1] = torch.zeros(0, 5)                          # the last image (last rank) has no face
    tgt = hp.ssd_encode_targets(boxes, (size, size))
    masks = SM.make_dropout_masks(fil, B, seed=43)
    lo, hi = rank * b_local, (rank + 1) * b_local
    if not live_dropout:
        model.set_dropout_masks({k: v[lo:hi] for k, v in masks.items()})
    xs, ys = x[lo:hi].cuda(), tgt[lo:hi].contiguous()
    losses = []
    for _ in range(steps):
        loss, _ = mm.fused_train_step(xs, ys)
        losses.append(float(loss))
    sp = mm.opt._space()
    return ({k: v.detach().cpu().clone() for k, v in model.state_dict().items()}, sp.grad.cpu().clone(), losses,
            mm._reducer is not None and mm._reducer.enabled)


if __name__ == "__main__":
    out, F_, b_local, steps, live = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # every rank starts from DIFFERENT weights: the start-of-training broadcast must put them on rank 0's
    ssd = len(sys.argv) > 6 and sys.argv[6] == "ssd"
    if ssd:
        params, grad, losses, enabled = run_rank_ssd(rank, world, b_local, steps, bool(live), param_seed=rank)
    else:
        params, grad, losses, enabled = run_rank(rank, world, F_, b_local, steps, bool(live), param_seed=rank)
    assert enabled, "the gradient reducer did not see the process group"
    tot = torch.tensor(losses, dtype=torch.float64)
    if not ssd:                                               # YOLO: per-rank loss sums add up; SSD: every rank holds the batch loss
        dist.all_reduce(tot)
    if rank == 0:
        torch.save({"params": params, "grad": grad, "loss_sum": tot.tolist(),
                    "grad_step1": None if ssd else run_rank.grad_step1}, out)
    dist.barrier()
    dist.destroy_process_group()
