"""MobileNetV3-small backbone (BASELINE.json config 5; models/MobilenetV3Backbone.py:11-60), bf16 inference on the GPU.

PARITY UNPINNED: no output of the reference exists for this model (timm absent, the TorchScript archive may not be
executed); the checker is oracle/mobilenet_oracle.py, an fp32 torch restatement of the architecture, run with the
archive's own parameters (tests/golden/g13_mobilenet_weights.npz).

Tolerances.  The HIP path keeps activations in bf16 (8 mantissa bits: relative rounding 2^-9 = 2e-3 per stored tensor) and
accumulates in fp32.  Per-kernel tests feed the torch reference the SAME bf16-rounded inputs and weights, so the only
difference left is the rounding of the output: |err| <= 2^-8 * |value| + small absolute slack.  The whole network stacks
~35 roundings; its bound is stated at the test.
"""
import warnings

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hp():
    import fdet_amd
    from fdet_amd import hotpath
    return hotpath


def _bf(t):
    return t.to(torch.bfloat16).float()


def _close_bf16(got, want, what, slack=2e-3):
    got, want = got.float().cpu(), want.float().cpu()
    err = (got - want).abs()
    bound = want.abs() * 2.0 ** -8 + slack
    bad = err > bound
    assert not bool(bad.any()), f"{what}: {int(bad.sum())} of {bad.numel()} outside bf16 rounding; max err {float(err.max())}"


def _act(x, a):
    return x if a == 0 else (F.relu(x) if a == 1 else F.hardswish(x))


@pytest.mark.parametrize("u8", [False, True])
def test_stem(hp, u8):
    g = torch.Generator().manual_seed(1)
    N, H, W = 3, 36, 44
    x = torch.randint(0, 256, (N, 3, H, W), generator=g, dtype=torch.uint8) if u8 else torch.rand(N, 3, H, W, generator=g)
    w = torch.randn(16, 3, 3, 3, generator=g) * 0.3
    b = torch.randn(16, generator=g) * 0.1
    y = hp.mb_stem(x.cuda(), w.reshape(16, 27).cuda(), b.cuda())
    xf = x.float() / 255.0 if u8 else x
    want = F.hardswish(F.conv2d(F.pad(xf, [0, 1, 0, 1]), w, b, 2))                  # TF SAME on even sizes: 0 front, 1 back
    assert tuple(y.shape) == (N, H // 2, W // 2, 16)
    _close_bf16(y.permute(0, 3, 1, 2), want, "stem")


@pytest.mark.parametrize("C,K,stride,H,act,pool", [(16, 3, 2, 24, 1, True), (72, 3, 2, 12, 1, False), (88, 3, 1, 9, 1, False),
                                                   (96, 5, 2, 14, 2, True), (240, 5, 1, 7, 2, True), (576, 5, 1, 5, 2, True),
                                                   (288, 5, 2, 10, 2, True), (24, 5, 1, 33, 0, True)])
def test_depthwise_and_se_pool(hp, C, K, stride, H, act, pool):
    g = torch.Generator().manual_seed(C + K)
    N = 3
    x = _bf(torch.randn(N, C, H, H, generator=g))
    w = torch.randn(C, 1, K, K, generator=g) * 0.3
    b = torch.randn(C, generator=g) * 0.1
    y, ps = hp.mb_depthwise(x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).cuda(), w.reshape(C, K * K).t().contiguous().cuda(),
                            b.cuda(), K, stride, act, pool)
    if stride == 1:
        want = F.conv2d(x, w, b, 1, K // 2, 1, C)
    else:
        Ho = -(-H // 2)
        tot = max((Ho - 1) * 2 + K - H, 0)
        want = F.conv2d(F.pad(x, [tot // 2, tot - tot // 2] * 2), w, b, 2, 0, 1, C)
    want = _act(want, act)
    _close_bf16(y.permute(0, 3, 1, 2), want, "depthwise")
    if pool:
        sums = y.float().sum((1, 2)).cpu()                       # the pool sums the ROUNDED outputs (what the next layer reads)
        got = ps.sum(1).cpu()                                    # (N, slots, C): one partial row per workgroup column
        assert torch.allclose(got, sums, rtol=1e-4, atol=1e-3), float((got - sums).abs().max())
    else:
        assert ps is None


@pytest.mark.parametrize("C,R,HW", [(16, 8, 120 * 120), (96, 24, 900), (576, 144, 225)])
def test_se_gate(hp, C, R, HW):
    g = torch.Generator().manual_seed(C)
    N = 5
    pool = torch.randn(N, C, generator=g) * HW * 0.3
    w1, b1 = torch.randn(R, C, generator=g) * 0.2, torch.randn(R, generator=g) * 0.1
    w2, b2 = torch.randn(C, R, generator=g) * 0.2, torch.randn(C, generator=g)
    gate = hp.mb_se_gate(pool.cuda(), HW, w1.cuda(), b1.cuda(), w2.cuda(), b2.cuda())
    want = F.hardsigmoid(F.linear(F.relu(F.linear(pool / HW, w1, b1)), w2, b2))
    assert torch.allclose(gate.cpu(), want, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("cin,cout,P,act,gate,res", [(16, 16, (13, 11), 0, True, False), (16, 72, (9, 9), 1, False, False),
                                                     (88, 24, (6, 7), 0, False, True), (240, 40, (5, 5), 0, True, True),
                                                     (40, 120, (15, 15), 2, False, False), (144, 48, (4, 33), 0, True, True),
                                                     (96, 576, (3, 3), 2, False, False), (576, 96, (15, 15), 0, True, True),
                                                     (24, 88, (1, 1), 1, False, False), (48, 288, (2, 129), 2, False, False)])
def test_pointwise(hp, cin, cout, P, act, gate, res):
    g = torch.Generator().manual_seed(cin * 7 + cout)
    N = 3
    x = _bf(torch.randn(N, P[0], P[1], cin, generator=g))
    w = torch.randn(cout, cin, generator=g) / cin ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    gt = torch.rand(N, cin, generator=g) if gate else None
    rs = _bf(torch.randn(N, P[0], P[1], cout, generator=g)) if res else None
    wp, bp = hp.mb_pointwise_pack(w.cuda(), b.cuda())
    y = hp.mb_pointwise(x.to(torch.bfloat16).cuda(), wp, bp, cout, act, gt.cuda() if gate else None,
                        rs.to(torch.bfloat16).cuda() if res else None)
    xin = _bf(x * gt[:, None, None, :]) if gate else x           # the gated operand is rounded to bf16 for the matrix cores
    want = _act(F.linear(xin, _bf(w), b), act)
    if res:
        want = want + rs
    _close_bf16(y, want, "pointwise", slack=4e-3)


def test_head(hp):
    g = torch.Generator().manual_seed(3)
    N, S, C = 3, 15, 576
    f = _bf(torch.randn(N, C, S, S, generator=g))
    w = torch.randn(5, C, 3, 3, generator=g) * 0.02
    b = torch.randn(5, generator=g)
    y = hp.mb_head(f.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).cuda(), hp.mb_head_pack(w.cuda()), b.cuda())
    want = torch.sigmoid(F.conv2d(f, w, b, padding=1))
    assert torch.allclose(y.cpu(), want, rtol=1e-4, atol=1e-5), float((y.cpu() - want).abs().max())


def _model(weights, size=480):
    from fdet_amd.models.MobilenetV3Backbone import MobilenetV3Backbone
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        net = MobilenetV3Backbone(64, (3, size, size), size // 32, pretrained=False)
    missing = net.load_state_dict(weights, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return net.cuda().eval()


def test_whole_model_against_oracle_with_archive_weights(golden):
    """(2,3,480,480) random images through the HIP bf16 stack vs the fp32 oracle with the SAME (archive) parameters.
    Bound: ~35 bf16 roundings of O(1) activations through a trained network, read through a sigmoid (slope <= 1/4):
    |map error| <= 0.03 everywhere and <= 4e-3 on average (observed values are printed by the test)."""
    from oracle import mobilenet_oracle as MO
    P = golden("g13_mobilenet_weights")
    net = _model(P)
    g = torch.Generator().manual_seed(5)
    x = torch.rand(2, 3, 480, 480, generator=g)
    x[1] = F.avg_pool2d(x[1:2], 9, 1, 4)[0]                      # one smooth image, one noise image
    got = net(x.cuda()).cpu()
    want = MO.model_forward({k: v for k, v in P.items()}, x)
    assert tuple(got.shape) == (2, 5, 15, 15)
    err = (got - want).abs()
    print("mobilenet whole-model |err| max", float(err.max()), "mean", float(err.mean()))
    assert float(err.max()) <= 0.03 and float(err.mean()) <= 4e-3


@pytest.mark.parametrize("weights", ["archive", "random"])
def test_every_block_against_oracle_on_the_same_input(golden, weights):
    """Layer-by-layer: each block of the HIP engine and the oracle's block are given the SAME input (the engine's own bf16
    activation), so the difference is what ONE block adds: its 2-3 stored bf16 tensors, the bf16 weight panels and the
    bf16 gated operand -- bounded by 1.5 % of the block output's RMS with the archive's (trained) parameters (observed
    0.7 %) and 4 % with random parameters (an ill-conditioned network: activations grow to RMS ~300 and the SqueezeExcite
    gates sit on their clipping edges; observed 2.2 %), far below what a wrong tap, pad, gate or residual gives (tens of
    percent)."""
    from oracle import mobilenet_oracle as MO
    P = golden("g13_mobilenet_weights") if weights == "archive" else MO.init_params(3)
    size = 160
    net = _model(P, size)
    eng = net._packed_engine()
    x = torch.rand(3, 3, size, size, generator=torch.Generator().manual_seed(size))
    h = eng.stem(x.cuda())
    _close_bf16(h.permute(0, 3, 1, 2), MO.stem_forward(P, x), "stem", slack=4e-3)
    worst, bound = 0.0, (0.015 if weights == "archive" else 0.04)
    for b in range(len(MO.BLOCKS)):
        hin = h.float().permute(0, 3, 1, 2).cpu()
        h = eng.block(b, h)
        want = MO.block_forward(P, b, hin)
        got = h.float().permute(0, 3, 1, 2).cpu()
        rel = float((got - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt())
        worst = max(worst, rel)
        assert rel <= bound, (b, rel)
    hin = h.float().permute(0, 3, 1, 2).cpu()
    f = eng.final(h).float().permute(0, 3, 1, 2).cpu()
    want = MO.final_forward(P, hin)
    rel = float((f - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt())
    print("mobilenet per-block relative rms error: worst", max(worst, rel))
    assert rel <= bound


def test_whole_backbone_features_accumulated_error(golden):
    """End to end with the archive parameters at 96x96 and 160x160: backbone FEATURES (before the head) against the fp32
    oracle; the ~35 roundings accumulate to a few percent of the feature RMS (bound 5 %)."""
    from oracle import mobilenet_oracle as MO
    P = golden("g13_mobilenet_weights")
    for size in (96, 160):
        net = _model(P, size)
        x = torch.rand(3, 3, size, size, generator=torch.Generator().manual_seed(size))
        f = net._packed_engine().features(x.cuda()).float().permute(0, 3, 1, 2).cpu()
        want = MO.features(P, x)
        rms = float(want.pow(2).mean().sqrt())
        err = float((f - want).pow(2).mean().sqrt())
        print("mobilenet features rms", rms, "rms err", err)
        assert err <= 0.05 * rms, (err, rms)


def test_boxes_against_oracle_away_from_the_threshold(golden):
    """Box level (parity UNPINNED: the fp32 oracle is a restatement, no reference output exists).  The bf16 stack's maps
    are within 0.03 of the oracle's (test above), so: (1) every cell whose oracle confidence is farther than 0.03 from the
    probability threshold is selected / rejected exactly as the oracle does; (2) the boxes decoded from cells selected by
    both (datasets/utils.py:107-162 arithmetic on either map) differ by 3 px on average, 99 % of the coordinates by at
    most 0.03 of the image size + 1 rounding pixel, none by more than a tenth of the image (images 2 and 3 are outside
    the [0,1] noise the 0.03 map bound was measured on; observed: mean 0.8 px, worst 33 px).  Cells inside the 0.03 band -- and hence NMS keep-sets that hinge on them -- may differ:
    that is the stated consequence of bf16 activations (inference only)."""
    from oracle import mobilenet_oracle as MO
    P = golden("g13_mobilenet_weights")
    net = _model(P)
    g = torch.Generator().manual_seed(17)
    x = torch.rand(4, 3, 480, 480, generator=g)
    x[1] = F.avg_pool2d(x[1:2], 9, 1, 4)[0]
    x[2] = F.avg_pool2d(x[2:3], 31, 1, 15)[0] * 1.5
    x[3, :, 100:300, 150:330] = x[3, :, 100:300, 150:330] * 0.3 + 0.5
    got = net(x.cuda()).cpu()
    want = MO.model_forward({k: v for k, v in P.items()}, x)
    S, size = 15, 480
    ps = size / S
    n_checked = n_boxes = 0
    worst = tot = 0.0
    alld = []
    for pt in (0.5, float(want[:, 0].quantile(0.9)), float(want[:, 0].quantile(0.5))):
        sure = (want[:, 0] - pt).abs() > 0.03
        sel_g, sel_w = got[:, 0] > pt, want[:, 0] > pt
        assert torch.equal(sel_g[sure], sel_w[sure])
        n_checked += int(sure.sum())
        both = sel_g & sel_w
        ii = torch.arange(S, dtype=torch.float32)[None, :, None].expand(4, S, S)
        jj = torch.arange(S, dtype=torch.float32)[None, None, :].expand(4, S, S)

        def boxes(m):
            X = m[:, 1] * ps + ii * ps
            Y = m[:, 2] * ps + jj * ps
            return torch.stack([X, Y, m[:, 3] * size + X, m[:, 4] * size + Y], 1).round()
        d = (boxes(got) - boxes(want)).abs().permute(0, 2, 3, 1)[both]
        if d.numel():
            n_boxes += d.shape[0]
            worst = max(worst, float(d.max()))
            tot += float(d.sum()) / 4
            alld.append(d.reshape(-1))
    print("mobilenet box level: cells checked", n_checked, "boxes compared", n_boxes, "worst |d| px", worst,
          "mean px", tot / max(n_boxes, 1))
    assert n_checked > 1000 and n_boxes > 50
    q99 = float(torch.cat(alld).quantile(0.99))
    print("   99th percentile px", q99)
    assert q99 <= 0.03 * size + 1 and worst <= 0.1 * size and tot / max(n_boxes, 1) <= 3.0


def test_predict_path_uint8_frames(golden):
    """forward(frames, predict=1): uint8 frames at the model size go straight to the stem (the /255 fused); the result
    equals decode+NMS of the maps from the float path, and frames of another size go through the resize kernel."""
    P = golden("g13_mobilenet_weights")
    net = _model(P)
    g = torch.Generator().manual_seed(9)
    frame = torch.randint(0, 256, (3, 480, 480), generator=g, dtype=torch.uint8)
    boxes = net(frame.cuda(), torch.tensor(1))
    maps = net((frame.float() / 255.0)[None].cuda())
    want = net.single_non_max_suppression(maps[0])
    assert boxes.shape[1] == 5 and torch.equal(boxes.cpu(), want.cpu())
    other = torch.randint(0, 256, (3, 300, 400), generator=g, dtype=torch.uint8)
    b2 = net(other.cuda(), torch.tensor(1))
    assert b2.dim() == 2 and b2.shape[1] == 5


def test_forward_is_bit_reproducible(golden):
    """No float atomics anywhere in the forward: two runs of the same batch (and a run inside a larger batch) give
    bit-identical maps."""
    net = _model(golden("g13_mobilenet_weights"))
    x = torch.rand(5, 3, 480, 480, generator=torch.Generator().manual_seed(21)).cuda()
    a = net(x)
    for _ in range(3):
        assert torch.equal(net(x), a)
    assert torch.equal(net(x[:2].contiguous()), a[:2])


def test_graphed_predict_equals_eager(golden):
    """The launch-bound demo path (preprocess -> ~45 backbone launches -> decode -> NMS) replayed from ONE HIP graph gives
    the boxes of the eager path, frame after frame."""
    net = _model(golden("g13_mobilenet_weights"))
    net.reduce_bounding_boxes.probability_threshold = 0.3
    g = torch.Generator().manual_seed(11)
    frames = [torch.randint(0, 256, (2, 3, 480, 480), generator=g, dtype=torch.uint8).cuda() for _ in range(3)]
    gp = net.graphed_predict(frames[0])
    for f in frames:
        want = net(f, torch.tensor(1))
        got = gp.first(f)
        assert torch.equal(got.cpu(), want.cpu())


def test_torchscript_export_matches_eager(golden, tmp_path):
    """`model.to_torchscript(path)` (the `LightningModule.to_torchscript` contract, train_model.py:61) for the MobileNetV3
    mirror: the scripted module (operator `fdet::mobilenet_forward`) gives the eager maps and the eager demo-path boxes
    bit for bit, before and after a torch.jit.save / load round trip."""
    net = _model(golden("g13_mobilenet_weights"))
    net.reduce_bounding_boxes.probability_threshold = 0.3
    path = tmp_path / "mobilenet_scripted.pt"
    scripted = net.to_torchscript(str(path))
    loaded = torch.jit.load(str(path))
    g = torch.Generator().manual_seed(31)
    x = torch.rand(2, 3, 480, 480, generator=g).cuda()
    frames = torch.randint(0, 256, (2, 3, 480, 480), generator=g, dtype=torch.uint8).cuda()
    other = torch.randint(0, 256, (3, 300, 400), generator=g, dtype=torch.uint8).cuda()
    with torch.no_grad():
        y_e = net(x)
        det_e = net(frames, torch.tensor(1))
        det_o = net(other, torch.tensor(1))
        for sm in (scripted, loaded):
            assert torch.equal(sm(x), y_e)
            det_s = sm(frames, torch.tensor(1))
            assert det_s.shape == det_e.shape and torch.equal(det_s.cpu(), det_e.cpu())
            det_t = sm(other, torch.tensor(1))
            assert det_t.shape == det_o.shape and torch.equal(det_t.cpu(), det_o.cpu())


def test_cpu_inputs_fail_loudly(golden):
    from fdet_amd import _native as N
    net = _model(golden("g13_mobilenet_weights"))
    with pytest.raises(N.FdetError):
        net(torch.rand(1, 3, 480, 480))
    net.train()
    with pytest.raises(N.FdetError):                       # training mode too: no CPU fallback
        net(torch.rand(1, 3, 480, 480))
