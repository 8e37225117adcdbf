"""CPU-only checks of the host-side mirror: geometry, parameter naming, optimizer plumbing."""
import pytest
import torch

import fdet_amd
from fdet_amd.convstack import StackGeometry, param_names
from fdet_amd.models.PoolResnet import PoolResnet
from fdet_amd.models.Resnet import Resnet


def test_state_dict_names_and_param_counts_match_reference():
    m = PoolResnet(64, (3, 480, 480), 10)
    assert sum(p.numel() for p in m.parameters()) == 769349          # SURVEY.md section 0 ("medium")
    assert sum(p.numel() for p in PoolResnet(32, (3, 480, 480), 10).parameters()) == 200357
    assert sum(p.numel() for p in PoolResnet(128, (3, 480, 480), 10).parameters()) == 3013253
    assert sum(p.numel() for p in Resnet(64, (3, 480, 480), 15).parameters()) == 743237
    keys = list(m.state_dict().keys())
    assert keys == param_names(10)
    assert keys[0] == "conv1.weight" and keys[-1] == "out.bias" and "residual_blocks.9.conv2.bias" in keys
    assert tuple(m.state_dict()["conv1.weight"].shape) == (64, 3, 10, 10)
    assert tuple(m.state_dict()["out.weight"].shape) == (5, 64, 6, 6)


def test_geometry_pool_rules():
    h0, lv = PoolResnet(64, (3, 480, 480), 10).engine.h0, PoolResnet(64, (3, 480, 480), 10).engine.lv
    assert h0 == 60 and lv[:3] == [(60, 2), (30, 2), (15, 1)]             # pool iff H > 2S (PoolResnet.py:41)
    lv = Resnet(64, (3, 640, 640), 20).engine.lv
    assert [h for h, _ in lv[:5]] == [320, 160, 80, 40, 20]                # pool iff H > S (Resnet.py:38)
    with pytest.raises(ValueError):                                        # 640^2 cannot reach a 15x15 grid
        Resnet(64, (3, 640, 640), 16).engine
    with pytest.raises(AssertionError):                                    # BaseModel.py:23-26
        PoolResnet(8, (3, 481, 481), 10)


def test_default_init_equals_reference_init_order():
    """Same seed -> same values as the reference constructor (Conv2d default init, module order)."""
    import oracle as O
    torch.manual_seed(0)
    m = PoolResnet(8, (3, 480, 480), 10)
    P = O.init_params(O.poolresnet_spec(8, (3, 480, 480), 10), seed=0)
    for k, v in m.state_dict().items():
        assert torch.equal(v, P[k]), k


def test_lightning_style_checkpoint_round_trip(tmp_path):
    """the reference restores training state with
    `ModelMeta(model=model).load_state_dict(checkpoint["state_dict"])` (demo_scripts/convert_checkpoint_to_scripted_model.py:32-40,
    load_checkpoint.py:22) -- Lightning keys carry the `model.` prefix.  The mirror's ModelMeta has the same key set, and a
    checkpoint written in that layout loads back through a loader that executes nothing from the file."""
    from fdet_amd.models import ModelMeta
    torch.manual_seed(1)
    src = ModelMeta(model=PoolResnet(8, (3, 480, 480), 10), lr=1e-4)
    keys = list(src.state_dict().keys())
    assert keys == ["model." + k for k in param_names(10)]
    path = tmp_path / "epoch=0-step=1.ckpt"
    torch.save({"state_dict": src.state_dict(), "epoch": 0, "global_step": 1}, path)
    ck = torch.load(path, map_location="cpu", weights_only=True)
    torch.manual_seed(2)
    dst = ModelMeta(model=PoolResnet(8, (3, 480, 480), 10), lr=1e-4)
    assert not torch.equal(dst.state_dict()["model.conv1.weight"], src.state_dict()["model.conv1.weight"])
    res = dst.load_state_dict(ck["state_dict"])
    assert not res.missing_keys and not res.unexpected_keys
    for k in keys:
        assert torch.equal(dst.state_dict()[k], src.state_dict()[k]), k
    # the bare model takes the same tensors with the prefix stripped (how the demos hand weights to BaseModel)
    bare = PoolResnet(8, (3, 480, 480), 10)
    bare.load_state_dict({k[len("model."):]: v for k, v in ck["state_dict"].items()})
    assert torch.equal(bare.state_dict()["out.bias"], src.state_dict()["model.out.bias"])


def test_product_path_has_no_cpu_fallback():
    from fdet_amd import FdetError
    m = PoolResnet(8, (3, 480, 480), 10).eval()
    with pytest.raises(FdetError):
        m(torch.rand(1, 3, 480, 480))


def test_product_never_imports_oracle():
    import os, re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                        "pytorch-face-detection-from-scratch_amd")
    for dp, _, fs in os.walk(root):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", src, re.M), os.path.join(dp, f)


def test_optimizer_state_dict_round_trip_in_adam_layout():
    """SAMSGD exports / loads its flat Adam moments in torch.optim.Adam's per-parameter layout, so a
    resumed run keeps its moments and bias-correction step, and the reference's `optimizer_states` load."""
    import fdet_amd  # noqa: F401
    from fdet_amd.optim import SAMSGD
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(3, 5)), torch.nn.Parameter(torch.randn(7)), torch.nn.Parameter(torch.randn(2, 2, 3, 3))]
    opt = SAMSGD(ps, lr=1e-3)
    sp = opt._space()
    sp.exp_avg.uniform_(-1, 1); sp.exp_avg_sq.uniform_(0, 1); opt.step_count = 7
    sd = opt.state_dict()
    assert sorted(sd["state"]) == [0, 1, 2] and float(sd["state"][1]["step"]) == 7.0
    for i, p in enumerate(ps):
        assert sd["state"][i]["exp_avg"].shape == p.shape
    # (1) into a fresh SAMSGD
    ps2 = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    opt2 = SAMSGD(ps2, lr=5e-2)
    opt2.load_state_dict(sd)
    assert opt2.step_count == 7 and opt2.param_groups[0]["lr"] == 1e-3
    for i in range(3):
        assert torch.equal(opt2.space.view(opt2.space.exp_avg, i), sp.view(sp.exp_avg, i))
        assert torch.equal(opt2.space.view(opt2.space.exp_avg_sq, i), sp.view(sp.exp_avg_sq, i))
    # (2) torch.optim.Adam reads the same dict (the reference's SAMSGD IS an Adam subclass) ...
    ref = torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in ps], lr=1e-3)
    ref.load_state_dict({"state": sd["state"], "param_groups": ref.state_dict()["param_groups"]})
    rp = ref.param_groups[0]["params"]
    assert torch.equal(ref.state[rp[2]]["exp_avg_sq"], sp.view(sp.exp_avg_sq, 2))
    # ... and a stock Adam checkpoint loads into SAMSGD
    for p in rp:
        p.grad = torch.randn_like(p)
    ref.step()
    opt3 = SAMSGD([torch.nn.Parameter(p.detach().clone()) for p in ps], lr=1e-3)
    opt3.load_state_dict(ref.state_dict())
    assert opt3.step_count == 8
    assert torch.equal(opt3.space.view(opt3.space.exp_avg, 0), ref.state[rp[0]]["exp_avg"])
    # before the first step there is no state to export, as torch's optimizers
    assert SAMSGD([torch.nn.Parameter(torch.zeros(3))], lr=1e-3).state_dict()["state"] == {}


def test_optimizer_step_calls_the_closure_with_grad_enabled():
    import fdet_amd  # noqa: F401
    from fdet_amd.optim import SAMSGD
    p = torch.nn.Parameter(torch.ones(4))
    opt = SAMSGD([p], lr=1e-3)
    seen = {}

    def closure():
        seen["grad_enabled"] = torch.is_grad_enabled()
        loss = (p * 2).sum()
        loss.backward()
        return loss

    opt._step = lambda *a, **k: seen.setdefault("stepped", p.grad is not None)      # no GPU here: stub the launch
    with torch.no_grad():
        out = opt.step(closure)
    assert seen == {"grad_enabled": True, "stepped": True} and float(out) == 8.0


def test_dropout_call_ranges_are_disjoint_for_any_batch_size():
    """the counter base of call k must not depend on the batch size."""
    import fdet_amd  # noqa: F401
    from fdet_amd.dataparallel import dropout_stream
    spans = []
    for calls, n in ((1, 256), (2, 3), (3, 1024), (4, 1)):
        base, first = dropout_stream(calls, n)
        assert first == 0
        spans.append((base, base + (first + n) * 11 * 64))
    for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
        assert a1 <= b0


def test_package_synthetic_boxes_equal_the_oracles():
    import fdet_amd  # noqa: F401
    import oracle as O
    from fdet_amd.datasets.synthetic import synthetic_boxes
    for a, b in zip(synthetic_boxes(16, 480, seed=101), O.synthetic_boxes(16, 480, seed=101)):
        assert torch.equal(a, b)


def test_bench_self_launch_refuses_more_ranks_than_gpus_without_touching_one():
    """`bench.py --gpus 2` with no launcher environment starts rank processes itself; on a box with fewer GPUs it
    says so (exit 2) instead of silently running one rank and printing n_gpus: 1 (VERDICT r1 weak #8a)."""
    import os
    import subprocess
    import sys
    from conftest import REPO
    if torch.cuda.device_count() >= 2:
        pytest.skip("box has >= 2 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "FDET_SINGLE_DEVICE")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 2 and "GPU(s) are visible" in r.stderr and r.stdout.strip() == ""


def test_torchscript_export_scripts_and_round_trips_without_a_gpu(tmp_path):
    """the inference path scripts (custom fdet:: operators), keeps the reference's state-dict names,
    survives torch.jit.save / load, and -- like every op of this package -- refuses CPU tensors loudly."""
    import fdet_amd
    from fdet_amd.models import ModelMeta
    m = PoolResnet(16, (3, 480, 480), 10)
    mm = ModelMeta(model=m)
    path = tmp_path / "scripted.pt"
    sm = mm.to_torchscript(str(path))
    kinds = [n.kind() for n in sm.graph.nodes()] + [n.kind() for b in sm.graph.nodes() for blk in b.blocks() for n in blk.nodes()]
    assert "fdet::stack_forward" in kinds and "fdet::preprocess" in kinds and "fdet::reduce_bounding_boxes" in kinds
    assert list(sm.state_dict().keys()) == param_names(10)
    loaded = torch.jit.load(str(path))
    assert list(loaded.state_dict().keys()) == param_names(10)
    for k, v in m.state_dict().items():
        assert torch.equal(loaded.state_dict()[k], v)
    with pytest.raises(Exception, match="GPU only"):
        loaded(torch.zeros(1, 3, 480, 480))
    # Resnet (3x3 head, other pool rule) scripts too
    r = Resnet(16, (3, 240, 240), 15, num_of_residual_blocks=6)
    assert "fdet::stack_forward" in [n.kind() for n in r.to_torchscript().graph.nodes()]


def test_mobilenet_mirror_state_dict_names_and_loud_failures():
    """models/MobilenetV3Backbone.py:11-60 mirror: same ctor signature, the state-dict names/shapes of the reference's
    module tree (so its checkpoints load by name), and no CPU fallback."""
    import inspect
    import warnings
    from fdet_amd import _native as N
    from fdet_amd.models.MobilenetV3Backbone import MobilenetV3Backbone
    from oracle import mobilenet_oracle as MO
    sig = list(inspect.signature(MobilenetV3Backbone.__init__).parameters)
    assert sig == ["self", "filters", "input_shape", "num_of_patches", "probability_threshold", "iou_threshold", "pretrained",
                   "input_kernel_size", "input_stride", "output_kernel_size", "output_padding"]
    with pytest.warns(UserWarning):
        MobilenetV3Backbone(64, (3, 480, 480), 15)               # pretrained=True: no download is attempted, says so
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        net = MobilenetV3Backbone(64, (3, 480, 480), 15, pretrained=False).eval()
    sd = net.state_dict()
    assert list(sd.keys()) == MO.param_names()
    assert all(tuple(sd[n].shape) == tuple(s) for n, s in MO.param_shapes().items())
    with pytest.raises(N.FdetError):
        net(torch.rand(1, 3, 480, 480))                          # CPU tensors: loud, no fallback
    with pytest.raises(ValueError):
        MobilenetV3Backbone(64, (3, 480, 480), 10, pretrained=False)      # stride 32 gives a 15x15 grid
