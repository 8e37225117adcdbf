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
