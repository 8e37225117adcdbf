"""One process of tests/test_gpu_rccl.py: three `fused_train_step`s of the product path, either with NO process group
(arg "0") or inside a ONE-rank RCCL group with the gradient exchange forced on (arg "1": FDET_DP_FORCE=1 -- the
all-reduce of one rank is the identity, so the parameters must come out bit-identical).  What the second form exercises
against RCCL instead of gloo: init_process_group("nccl", device_id=...), the parameter broadcast + checksum, async
all-reduce on slice views of the flat gradient launched from `after_block` while backward is still enqueueing kernels,
the stream-ordered wait() before Adam, and (SSD) the three-scalar exchange before backward.

    python tests/rccl_worker.py yolo|ssd 0|1 out.pt
"""
import json
import os
import socket
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def main():
    kind, use_dp, out = sys.argv[1], sys.argv[2] == "1", sys.argv[3]
    if use_dp:
        os.environ["FDET_DP_FORCE"] = "1"
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    info = {"kind": kind, "dp": use_dp}
    if use_dp:
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
        info["backend"] = dist.get_backend()
        info["world"] = dist.get_world_size()
    import fdet_amd
    from fdet_amd import hotpath as hp
    from fdet_amd.datasets.synthetic import synthetic_boxes
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(5)
    if kind == "yolo":
        from fdet_amd.models import ModelMeta
        from fdet_amd.models.PoolResnet import PoolResnet
        B, size, S = 4, 480, 10
        model = PoolResnet(filters=64, input_shape=(3, size, size), num_of_patches=S, num_of_residual_blocks=10).to(dev).train()
        mm = ModelMeta(model=model, lr=1e-3)
        x = torch.rand(B, 3, size, size, generator=g).to(dev)
        y = hp.encode_targets(synthetic_boxes(B, size, seed=6), (size, size), S, device=dev)
    else:
        from fdet_amd.models.ModelMetaSSD import ModelMetaSSD
        from fdet_amd.models.SSD import SSD
        B, size = 2, 480
        model = SSD(filters=16, input_shape=(3, size, size)).to(dev).train()
        mm = ModelMetaSSD(model=model, lr=1e-3)
        x = torch.rand(B, 3, size, size, generator=g).to(dev)
        y = hp.ssd_encode_targets(synthetic_boxes(B, size, seed=6), (size, size), device=dev)
    mm.configure_optimizers()
    losses = []
    for _ in range(3):
        r = mm.fused_train_step(x, y)
        losses.append(float(r[0]))
    red = mm._reducer
    info["reducer_enabled"] = bool(red is not None and red.enabled)
    info["losses"] = losses
    sp = mm.opt._space()
    torch.cuda.synchronize()
    torch.save({"flat": sp.flat.detach().cpu(), "info": info}, out)
    if use_dp:
        dist.destroy_process_group()
    print(json.dumps(info))


if __name__ == "__main__":
    main()
