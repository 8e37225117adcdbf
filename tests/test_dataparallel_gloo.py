"""N>1 path on CPU: world_size-2 `gloo` processes exercise the sharding + bucketed SUM
all-reduce of the flat gradient buffer (GradBucketReducer) and check the data-parallel oracle
of SURVEY.md 8(e): N-rank gradients == single-process gradients of the concatenated batch.
Per-rank gradients come from the CPU oracle (the HIP path cannot run here); what is under test
is the exchange logic that bench.py / ModelMeta.fused_train_step use unchanged on RCCL."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle as O


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out_q):
    import fdet_amd
    from fdet_amd.dataparallel import GradBucketReducer, shard_range, allreduce_scalars, sync_parameters, dropout_stream
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    F_, S, size, B = 8, 10, 480, 4
    spec = O.poolresnet_spec(F_, (3, size, size), S, 4, )
    P = O.init_params(spec, seed=0)
    g = torch.Generator().manual_seed(0)
    x = torch.rand(B, 3, size, size, generator=g)
    y = torch.stack([O.encode_targets(b, (size, size), S) for b in O.synthetic_boxes(B, size, seed=1)])
    masks = O.make_dropout_masks(spec, B, seed=2)
    lo, hi = shard_range(B, rank, world)
    names = list(P.keys())
    leaves = {k: P[k].clone().requires_grad_(True) for k in names}
    mk = {k: v[lo:hi] for k, v in masks.items()}
    loss = O.batch_loss(O.model_forward(spec, leaves, x[lo:hi], mk), y[lo:hi])
    grads = torch.autograd.grad(loss, [leaves[k] for k in names])
    # flat buffer in parameter order, 16-byte aligned views (as optim.FlatSpace lays it out)
    offs, off = [], 0
    for p in grads:
        offs.append(off); off += (p.numel() + 3) // 4 * 4
    flat = torch.zeros(off)
    for gr, o in zip(grads, offs):
        flat[o:o + gr.numel()] = gr.reshape(-1)
    split = offs[names.index("residual_blocks.2.conv1.weight")]
    red = GradBucketReducer(flat, split)
    assert red.enabled
    red.launch_tail()          # late bucket first (ready first in backward)
    red.launch_head()
    red.wait()
    tot = allreduce_scalars(loss.detach().reshape(1).clone())
    # start-of-training hand-shake: ranks that start from different weights continue from rank 0's
    w = torch.full((1000,), float(rank + 1)) + torch.arange(1000) * 1e-3
    sync_parameters(w)
    assert torch.equal(w, torch.full((1000,), 1.0) + torch.arange(1000) * 1e-3)
    # per-rank dropout streams: rank r owns the global images [r*n, (r+1)*n) of call k's counter range
    assert dropout_stream(3, 5) == (3 << 40, rank * 5)
    if rank == 0:
        out_q.put((flat, float(tot), offs, [tuple(g.shape) for g in grads], names))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_sum_allreduce_matches_single_process():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    flat, tot, offs, shapes, names = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference on the concatenated batch
    F_, S, size, B = 8, 10, 480, 4
    spec = O.poolresnet_spec(F_, (3, size, size), S, 4)
    P = O.init_params(spec, seed=0)
    g = torch.Generator().manual_seed(0)
    x = torch.rand(B, 3, size, size, generator=g)
    y = torch.stack([O.encode_targets(b, (size, size), S) for b in O.synthetic_boxes(B, size, seed=1)])
    masks = O.make_dropout_masks(spec, B, seed=2)
    leaves = {k: P[k].clone().requires_grad_(True) for k in names}
    loss = O.batch_loss(O.model_forward(spec, leaves, x, masks), y)
    grads = torch.autograd.grad(loss, [leaves[k] for k in names])
    assert abs(tot - float(loss)) <= 1e-5 * abs(float(loss))
    for gr, o, shp, n in zip(grads, offs, shapes, names):
        got = flat[o:o + gr.numel()].view(shp)
        assert torch.allclose(got, gr, rtol=1e-4, atol=1e-6 * float(gr.abs().max()) + 1e-9), n


def test_shard_range_partitions_the_batch():
    import fdet_amd
    from fdet_amd.dataparallel import shard_range
    for B in (1, 7, 8, 256, 257):
        for W in (1, 2, 3, 8):
            covered = []
            for r in range(W):
                lo, hi = shard_range(B, r, W)
                covered += list(range(lo, hi))
            assert covered == list(range(B))
