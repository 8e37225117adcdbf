"""End-to-end GPU parity of the model surface (PoolResnet / Resnet / ModelMeta) against the
golden fixtures produced by the reference and against the oracle.
Tolerances: fp32 outputs / loss within 1e-4 (north star); decoded boxes and keep-sets exact."""
import pytest
import torch

import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fd():
    import fdet_amd
    return fdet_amd


def _load(model, P):
    model.load_state_dict({k: v.clone() for k, v in P.items()})
    return model.cuda()


def _build(fd, kind, F, size, S, nb):
    from fdet_amd.models.PoolResnet import PoolResnet
    from fdet_amd.models.Resnet import Resnet
    cls = PoolResnet if kind == "poolresnet" else Resnet
    return cls(filters=F, input_shape=(3, size, size), num_of_patches=S, num_of_residual_blocks=nb)


def rel_close(a, b, tol=1e-4):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    scale = max(1e-30, float(b.abs().max()))
    err = float((a - b).abs().max())
    assert err <= tol * max(scale, 1e-3), f"max err {err:.3e}, scale {scale:.3e}"


@pytest.mark.parametrize("name,kind,size,S,nb", [("g5_poolresnet_F8", "poolresnet", 480, 10, 10),
                                                 ("g5_resnet_F8", "resnet", 240, 15, 6)])
def test_golden_train_step_autograd_path(fd, golden, name, kind, size, S, nb):
    """state_dict load -> eval forward -> train forward with the reference's dropout masks ->
    reference loss -> loss.backward() -> SAMSGD.step(), against the reference's numbers."""
    from fdet_amd.losses.YoloLoss import yolo_loss
    from fdet_amd.models import ModelMeta
    g = golden(name)
    P = {k[len("param/"):]: v for k, v in g.items() if k.startswith("param/")}
    masks = {k[len("mask/"):]: v for k, v in g.items() if k.startswith("mask/")}
    model = _load(_build(fd, kind, 8, size, S, nb), P)
    x = (g["x_u8"].float() / 255.0).cuda()
    model.eval()
    with torch.no_grad():
        y_eval = model(x)
    assert torch.allclose(y_eval.cpu(), g["y_eval"], atol=1e-4)
    mm = ModelMeta(model=model, lr=1e-4)
    (opt,), _ = mm.configure_optimizers()
    model.train()
    model.set_dropout_masks(masks)
    y_hat = mm(x)
    assert torch.allclose(y_hat.detach().cpu(), g["y_train"], atol=1e-4)
    y = g["y"].cuda()
    loss = 0
    for i in range(y.shape[0]):                       # the reference's per-image loop (ModelMeta.py:173-176)
        loss = loss + yolo_loss(y_hat[i], y[i])
    assert abs(float(loss) - float(g["loss"])) <= 1e-4 * max(1.0, float(g["loss"]))
    loss.backward()
    for n, p in model.named_parameters():
        rel_close(p.grad, g["grad/" + n])
    opt.step()
    for n, p in model.named_parameters():
        refp = g["param_after/" + n]
        d = (p.detach().cpu() - refp).abs()
        assert float(d.max()) <= 2.1e-4, n           # Adam's first step is ~lr*sign(g): sign noise at |g|~0
        assert float((d > 1e-6).float().mean()) < 0.02, n


@pytest.mark.parametrize("kind,F,size,S,nb,B", [("poolresnet", 64, 480, 10, 10, 3), ("poolresnet", 32, 480, 10, 10, 2),
                                                ("resnet", 16, 240, 15, 6, 2),
                                                # config-3 geometry (Resnet 640^2, S=20) scaled to 320^2: 160/80/40/20 maps
                                                ("resnet", 16, 320, 20, 5, 2)])
def test_fused_train_steps_vs_oracle(fd, kind, F, size, S, nb, B):
    """Three fused steps (forward + loss + backward + Adam, no autograd) against the oracle's
    train_step on the same inputs and dropout masks."""
    from fdet_amd.models import ModelMeta
    spec = (O.poolresnet_spec if kind == "poolresnet" else O.resnet_spec)(F, (3, size, size), S, nb)
    P = O.init_params(spec, seed=3)
    model = _load(_build(fd, kind, F, size, S, nb), P)
    mm = ModelMeta(model=model, lr=1e-4)
    mm.configure_optimizers()
    model.train()
    state = {"exp_avg": {k: torch.zeros_like(v) for k, v in P.items()},
             "exp_avg_sq": {k: torch.zeros_like(v) for k, v in P.items()}}
    g = torch.Generator().manual_seed(11)
    for step in range(1, 4):
        x = torch.rand(B, 3, size, size, generator=g)
        boxes = O.synthetic_boxes(B, size, seed=step)
        y = torch.stack([O.encode_targets(b, (size, size), S) for b in boxes])
        masks = O.make_dropout_masks(spec, B, seed=100 + step)
        loss_ref, y_ref, G_ref = O.train_step(spec, P, state, step, x, y, masks)
        model.set_dropout_masks(masks)
        lsum, y_hat, _ = mm.fused_train_step(x.cuda(), y.cuda())
        assert torch.allclose(y_hat.cpu(), y_ref, atol=1e-4)
        assert abs(float(lsum) - float(loss_ref)) <= 1e-4 * max(1.0, float(loss_ref))
        sp = mm.opt.space
        names, _ = model.named_stack_params()
        # step 1 starts from identical parameters: gradients within 2e-4 of each tensor's scale.
        # Adam's first update is ~lr*sign(g), so noise-level gradients can flip and later steps
        # start from parameters 2e-4 apart; the loss' 1/sqrt(p) terms amplify that: 2e-3 there.
        # The engine runs these channel counts in bf16x3 arithmetic (~1e-5 of fp32): forward and
        # loss stay within 1e-4 (asserted above).  The GRADIENT is a discontinuous function of the
        # activations (2x2 max-pool routing, LeakyReLU kinks): one tied window or one activation at a kink
        # that rounds the other way moves single entries by 1e-3..1e-2 of the tensor's scale -- in the
        # exact-fp32 arithmetic too.  test_gradient_entries_differ_only_through_pool_routing proves that
        # this is the ONLY source (per-entry 1e-4 once the oracle's decisions are injected); here, with
        # the kernel's own decisions, gradients are compared in the L2 sense with a loose per-entry bound.
        for i, n in enumerate(names):
            got = sp.view(sp.grad, i).detach().cpu().double()
            ref = G_ref[n].double()
            rel_l2 = float((got - ref).norm() / ref.norm().clamp_min(1e-30))
            assert rel_l2 <= (5e-3 if step == 1 else 5e-2), (n, step, rel_l2)
            rel_close(got, ref, 5e-2 if step == 1 else 2e-1)
    for n, p in model.named_parameters():
        d = (p.detach().cpu() - P[n]).abs()
        assert float(d.max()) <= 4.1e-4, n
        # sign flips of noise-level gradient entries under Adam (see above): a few per tensor at most
        assert int((d > 2e-6).sum()) <= max(3, int(0.03 * d.numel())), n


def test_trained_small_archive_demo_path(fd, golden):
    """demo_model.py path: uint8 frame stacked twice -> forward(predict=1) -> boxes of image 0,
    with the shipped small PoolResnet weights (thresholds 0.7 / 0.01)."""
    from fdet_amd.models.PoolResnet import PoolResnet
    g = golden("g6_trained_small")
    P = {k[len("param/"):]: v for k, v in g.items() if k.startswith("param/")}
    model = PoolResnet(filters=32, input_shape=(3, 480, 480), num_of_patches=10, probability_threshold=0.7,
                       iou_threshold=0.01)
    model = _load(model, P).eval()
    for n in range(g["images"].shape[0]):
        u8 = g["images"][n].cuda()
        with torch.no_grad():
            y = model(torch.stack([u8, u8]).float() / 255.0)
            det = model(torch.stack([u8, u8]), predict=torch.tensor(1))
        assert torch.allclose(y[0].cpu(), g["y"][n], atol=1e-4)
        nd = int(g["ndets"][n])
        assert det.shape == (nd, 5)
        if nd:
            assert torch.allclose(det[:, 0].cpu(), g["dets"][n, :nd, 0], atol=1e-4)
            assert torch.equal(det[:, 1:].cpu(), g["dets"][n, :nd, 1:])


def test_modelmeta_step_outputs_vs_oracle(fd):
    from fdet_amd.models import ModelMeta
    from fdet_amd.models.PoolResnet import PoolResnet
    spec = O.poolresnet_spec(16, (3, 480, 480), 10)
    P = O.init_params(spec, seed=5)
    P["out.bias"][0] += 1.0                                  # push confidences up so boxes appear
    model = _load(PoolResnet(16, (3, 480, 480), 10), P).eval()
    mm = ModelMeta(model=model)
    B = 6
    x = torch.rand(B, 3, 480, 480, generator=torch.Generator().manual_seed(1))
    boxes = O.synthetic_boxes(B, 480, seed=2)
    y = torch.stack([O.encode_targets(b, (480, 480), 10) for b in boxes])
    with torch.no_grad():
        out = mm.validation_step((x.cuda(), y.cuda(), boxes), 0)
        y_ref = O.model_forward(spec, P, x, None)
    loss_ref = O.batch_loss(y_ref, y)
    assert abs(float(out["loss"]) - float(loss_ref)) <= 1e-4 * float(loss_ref)
    iou, rec, prec = O.step_metrics(y_ref, y, O.ReduceBoundingBoxes(0.5, 0.5, (3, 480, 480), 10))
    assert abs(float(out["total_iou"]) - iou) <= 1e-4 * max(1.0, abs(iou))
    assert abs(float(out["total_recall"]) - rec) <= 1e-6
    assert abs(float(out["total_precision"]) - prec) <= 1e-6
    # non_max_suppression on a batch returns a tuple of per-image results (BaseModel.py:47-51)
    res = model.non_max_suppression(y.cuda())
    assert isinstance(res, tuple) and len(res) == B
    for n in range(B):
        ref = O.ReduceBoundingBoxes(0.5, 0.5, (3, 480, 480), 10)(y[n])
        assert torch.equal(res[n].cpu(), ref)


def test_basemodel_asserts_divisibility(fd):
    from fdet_amd.models.PoolResnet import PoolResnet
    with pytest.raises(AssertionError):
        PoolResnet(8, (3, 481, 481), 10)


def test_predict_resizes_frames_on_device(fd, golden):
    """forward(x, predict=1) with frames that are NOT at the model resolution: the bilinear Resize of
    models/PoolResnet.py:91,95 runs on the device (fdet_resize_bilinear_u8_norm).  Oracle:
    predict_image0 with torch CPU F.interpolate.  Trained small weights, thresholds 0.7 / 0.01."""
    from fdet_amd.models.PoolResnet import PoolResnet
    g = golden("g6_trained_small")
    P = {k[len("param/"):]: v for k, v in g.items() if k.startswith("param/")}
    spec = O.poolresnet_spec(32, (3, 480, 480), 10)
    model = PoolResnet(filters=32, input_shape=(3, 480, 480), num_of_patches=10, probability_threshold=0.7,
                       iou_threshold=0.01)
    model = _load(model, P).eval()
    for n in range(g["images"].shape[0]):
        u8 = g["images"][n]                                                    # (3,480,480) uint8
        big = u8.repeat_interleave(4, 1).repeat_interleave(4, 2)[:, 160:160 + 1280:2, 0:1920:3].contiguous()   # (3,640,640)
        assert tuple(big.shape) == (3, 640, 640)
        ref = O.predict_image0(spec, P, torch.stack([big, big]), 0.7, 0.01)
        with torch.no_grad():
            det = model(torch.stack([big, big]).cuda(), predict=torch.tensor(1))
        assert det.shape == ref.shape
        if ref.shape[0]:
            assert torch.allclose(det[:, 0].cpu(), ref[:, 0], atol=1e-4)
            assert torch.equal(det[:, 1:].cpu(), ref[:, 1:])


def test_u8_feeder_matches_direct_path(fd):
    """Pinned double-buffered uint8 feed (datasets/feed.py) delivers exactly what the direct
    `x.cuda()` + /255 path delivers, at the model resolution and through the on-device resize."""
    from fdet_amd.datasets.feed import U8BatchFeeder
    from fdet_amd import hotpath as hp
    g = torch.Generator().manual_seed(9)
    for shape in ((4, 3, 480, 480), (2, 3, 300, 400)):
        feeder = U8BatchFeeder(shape, (480, 480), "cuda", target_shape=(shape[0], 5, 10, 10), depth=2)
        batches = [(torch.randint(0, 256, shape, dtype=torch.uint8, generator=g), torch.rand(shape[0], 5, 10, 10, generator=g))
                   for _ in range(5)]
        feeder.submit(*batches[0])
        for i in range(len(batches)):
            if i + 1 < len(batches):
                feeder.submit(*batches[i + 1])
            x, y, tok = feeder.get()
            ref = hp.resize_bilinear_norm(batches[i][0].cuda(), (480, 480))
            assert torch.equal(x, ref)
            assert torch.equal(y.cpu(), batches[i][1])
            feeder.release(tok)


def test_graphed_predict_equals_eager(fd):
    """HIP-graph replay of the demo path returns exactly what forward(x, predict=1) returns, frame after frame."""
    from fdet_amd.models.PoolResnet import PoolResnet
    spec = O.poolresnet_spec(16, (3, 480, 480), 10)
    P = O.init_params(spec, seed=3)
    model = PoolResnet(16, (3, 480, 480), 10, probability_threshold=0.45, iou_threshold=0.5)
    model.load_state_dict({k: v.clone() for k, v in P.items()})
    model = model.cuda().eval()
    g = torch.Generator().manual_seed(5)
    frames = [torch.randint(0, 256, (2, 3, 480, 480), dtype=torch.uint8, generator=g).cuda() for _ in range(3)]
    gp = model.graphed_predict(frames[0])
    for f in frames + frames[:1]:
        eager = model(f, predict=torch.tensor(1))
        got = gp.first(f)
        assert got.shape == eager.shape and torch.equal(got.cpu(), eager.cpu())
        both = gp(f)
        assert len(both) == 2 and torch.equal(both[0].cpu(), eager.cpu())


def test_forward_frames_fuses_the_division_into_the_stem(fd):
    """forward_frames(uint8 frames) -- what forward(x, predict=1) and the batched serving path run -- hands the frames to the
    stem as they are (x / 255 in its staging, fdet_stem_fwd_ps_u8): the maps equal _stack_forward(_preprocess(frames)) bit for
    bit, in both precisions, and a float input / a training-mode model / a model without the pre-split path take the
    separate normalisation."""
    from fdet_amd.models.PoolResnet import PoolResnet
    spec = O.poolresnet_spec(64, (3, 480, 480), 10)
    P = O.init_params(spec, seed=3)
    model = PoolResnet(64, (3, 480, 480), 10)
    model.load_state_dict({k: v.clone() for k, v in P.items()})
    model = model.cuda().eval()
    assert model.engine.u8_frames_ok()
    frames = torch.randint(0, 256, (3, 3, 480, 480), dtype=torch.uint8, generator=torch.Generator().manual_seed(8)).cuda()
    with torch.no_grad():
        for mode in ("bf16x3", "bf16"):
            model.engine.set_precision(mode)
            two_step = model._stack_forward(model._preprocess(frames))
            fused = model.forward_frames(frames)
            assert torch.equal(fused, two_step), mode
        model.engine.set_precision("bf16x3")
        assert torch.equal(model.forward_frames(frames.float()), model._stack_forward(frames.float() / 255.0))
    small = PoolResnet(16, (3, 480, 480), 10).cuda().eval()      # 16 filters: no pre-split path, no fused stem
    assert not small.engine.u8_frames_ok()
    with torch.no_grad():
        assert torch.equal(small.forward_frames(frames), small._stack_forward(small._preprocess(frames)))


def _redraw_u8(B, size, seed, checksum):
    x_u8 = torch.randint(0, 256, (B, 3, size, size), generator=torch.Generator().manual_seed(seed), dtype=torch.uint8)
    assert int(x_u8.long().sum()) == int(checksum)
    return x_u8


def test_reference_modelmeta_two_steps_lightning_path(fd, golden):
    """g11 = the REFERENCE's ModelMeta.training_step + loss.backward() + SAMSGD.step(), twice (tools/make_goldens_r2.py).
    The mirror goes the same way (training_step -> loss.backward() -> optimizer.step()): loss, metric block and the
    parameters after each step against the reference's.  Q18 (the SAM wrapper's w+e-e rounding, <= 7.5e-9 in the fixture)
    is below the comparison's resolution."""
    from fdet_amd.models import ModelMeta
    g = golden("g11_modelmeta_F8")
    B, steps, seed = int(g["B"]), int(g["steps"]), int(g["seed"])
    P0 = {k[len("param0/"):]: v for k, v in g.items() if k.startswith("param0/")}
    model = _load(_build(fd, "poolresnet", 8, 480, 10, 10), P0).train()
    mm = ModelMeta(model=model, lr=1e-4)
    (opt,), _ = mm.configure_optimizers()
    for st in range(1, steps + 1):
        x = (_redraw_u8(B, 480, seed + st, g[f"x_checksum/{st}"]).float() / 255.0).cuda()
        y = g[f"y/{st}"].cuda()
        boxes = [g[f"boxes/{st}/{i}"] for i in range(B)]
        model.set_dropout_masks({k[len(f"mask/{st}/"):]: v for k, v in g.items() if k.startswith(f"mask/{st}/")})
        out = mm.training_step((x, y, boxes), 1)
        assert abs(float(out["loss"]) - float(g[f"loss/{st}"])) <= 1e-4 * float(g[f"loss/{st}"])
        m = g[f"metrics/{st}"]
        assert abs(float(out["total_iou"]) - float(m[0])) <= 1e-4 * max(1.0, float(m[0]))
        assert abs(float(out["total_recall"]) - float(m[1])) <= 1e-6 and abs(float(out["total_precision"]) - float(m[2])) <= 1e-6
        opt.zero_grad()
        out["loss"].backward()
        for n, p in model.named_parameters():
            rel_close(p.grad, g[f"grad/{st}/{n}"], 1e-4 if st == 1 else 2e-3)   # step 2 starts ~lr*sign-noise apart
        opt.step()
        for n, p in model.named_parameters():
            d = (p.detach().cpu() - g[f"param/{st}/{n}"]).abs()
            assert float(d.max()) <= 2.1e-4 * st, (st, n)
            assert float((d > 1e-6).float().mean()) < 0.02, (st, n)


def test_reference_f16_train_step_reaches_bf16x3_backward(fd, golden):
    """g12 = one reference train step of PoolResnet(filters=16): the reference-generated fixture that runs through
    the bf16x3 forward, data-gradient and weight-gradient kernels (16 channels is their narrowest width), the pooled
    blocks' fused epilogues included."""
    from fdet_amd.models import ModelMeta
    g = golden("g12_poolresnet_F16")
    P = {k[len("param/"):]: v for k, v in g.items() if k.startswith("param/")}
    model = _load(_build(fd, "poolresnet", 16, 480, 10, 10), P).train()
    assert model.engine.x3
    mm = ModelMeta(model=model, lr=1e-4)
    mm.configure_optimizers()
    x = (_redraw_u8(int(g["B"]), 480, int(g["seed"]), g["x_checksum"]).float() / 255.0).cuda()
    model.set_dropout_masks({k[len("mask/"):]: v for k, v in g.items() if k.startswith("mask/")})
    lsum, y_hat, _ = mm.fused_train_step(x, g["y"].cuda())
    assert torch.allclose(y_hat.cpu(), g["y_train"], atol=1e-4)
    assert abs(float(lsum) - float(g["loss"])) <= 1e-4 * max(1.0, float(g["loss"]))
    sp = mm.opt.space
    names, _ = model.named_stack_params()
    for i, n in enumerate(names):
        got = sp.view(sp.grad, i).detach().cpu().double()
        ref = g["grad/" + n].double()
        rel_l2 = float((got - ref).norm() / ref.norm().clamp_min(1e-30))
        assert rel_l2 <= 2e-3, (n, rel_l2)
        rel_close(got, ref, 2e-2)                     # single entries: max-pool routing of near-tied windows, see
        #                                               test_gradient_entries_differ_only_through_pool_routing
    for n, p in model.named_parameters():
        d = (p.detach().cpu() - g["param_after/" + n]).abs()
        assert float(d.max()) <= 2.1e-4, n
        assert float((d > 1e-6).float().mean()) < 0.02, n


def test_reference_f64_train_step_through_the_pre_split_kernels(fd, golden):
    """g15 = one train step of the REFERENCE PoolResnet at the headline width (filters 64, B=2; tools/make_goldens_r4.py):
    the reference-produced numbers that reach `k_conv3x3_ps`, `k_wgrad3x3_ps`, `k_block_chain_ps` -- the whole timed path
    (the engine must be on its PS path).  Forward and loss at 1e-4 (north star); every gradient tensor in the L2 sense and
    per entry on the fixture's 4096-entry sample; parameters after Adam."""
    from fdet_amd.models import ModelMeta
    from fdet_amd.models.PoolResnet import PoolResnet
    g = golden("g15_poolresnet_F64")
    torch.manual_seed(int(g["param_seed"]))
    model = PoolResnet(filters=64, input_shape=(3, 480, 480), num_of_patches=10, num_of_residual_blocks=10)
    names_ref = [str(n) for n in g["names"]]
    sd = model.state_dict()
    assert list(sd.keys()) == names_ref
    for n, s in zip(names_ref, g["param_sum"].tolist()):
        assert float(sd[n].double().sum()) == s, n          # same seed -> the reference constructor's parameters
    model = model.cuda().train()
    eng = model.engine
    assert eng.x3 and eng.ps and eng._ps_block(0) and eng._ps_block(1) and eng._ps_chain(2)
    mm = ModelMeta(model=model, lr=1e-4)
    mm.configure_optimizers()
    x_u8 = _redraw_u8(2, 480, int(g["x_seed"]), g["x_sum"])
    assert torch.equal(x_u8[:, :, ::97, ::89], g["x_probe"])
    model.set_dropout_masks({k[len("mask/"):]: v for k, v in g.items() if k.startswith("mask/")})
    lsum, y_hat, _ = mm.fused_train_step((x_u8.float() / 255.0).cuda(), g["y"].cuda())
    assert torch.allclose(y_hat.cpu(), g["y_train"], atol=1e-4)
    assert abs(float(lsum) - float(g["loss"])) <= 1e-4 * max(1.0, float(g["loss"]))
    sp = mm.opt.space
    names, _ = model.named_stack_params()
    assert list(names) == names_ref
    for i, n in enumerate(names):
        got = sp.view(sp.grad, i).detach().cpu().double().reshape(-1)
        idx = g["idx/" + n].long()
        ref = g["grad/" + n].double()
        scale = float(g["grad_absmax"][i])
        # whole-tensor check through the stored norm; the sample bounds single entries (max-pool routing of near-tied
        # windows moves single entries, see test_gradient_entries_differ_only_through_pool_routing)
        assert abs(float(got.norm()) - float(g["grad_norm"][i])) <= 2e-3 * float(g["grad_norm"][i]), n
        rel_l2 = float((got[idx] - ref).norm() / ref.norm().clamp_min(1e-30))
        assert rel_l2 <= 2e-3, (n, rel_l2)
        assert float((got[idx] - ref).abs().max()) <= 2e-2 * scale, n
    for n, p in model.named_parameters():
        d = (p.detach().cpu().reshape(-1)[g["idx/" + n].long()] - g["param_after/" + n]).abs()
        assert float(d.max()) <= 2.1e-4, n
        assert float((d > 1e-6).float().mean()) < 0.02, n


def test_trained_medium_archive_demo_path(fd, golden):
    """g16: demo_model.py:11-21 with the shipped MEDIUM PoolResnet weights (filters 64: the PS inference kernels), thresholds
    0.7 / 0.01: conv-stack output at 1e-4, boxes of image 0 exact."""
    from fdet_amd.models.PoolResnet import PoolResnet
    g = golden("g16_trained_medium")
    images = golden("g6_trained_small")["images"]
    P = {k[len("param/"):]: v for k, v in g.items() if k.startswith("param/")}
    model = PoolResnet(filters=64, input_shape=(3, 480, 480), num_of_patches=10, probability_threshold=0.7, iou_threshold=0.01)
    model = _load(model, P).eval()
    assert model.engine.ps
    for n in range(images.shape[0]):
        u8 = images[n].cuda()
        with torch.no_grad():
            y = model(torch.stack([u8, u8]).float() / 255.0)
            det = model(torch.stack([u8, u8]), predict=torch.tensor(1))
        assert torch.allclose(y[0].cpu(), g["y"][n], atol=1e-4)
        nd = int(g["ndets"][n])
        assert det.shape == (nd, 5)
        if nd:
            assert torch.allclose(det[:, 0].cpu(), g["dets"][n, :nd, 0], atol=1e-4)
            assert torch.equal(det[:, 1:].cpu(), g["dets"][n, :nd, 1:])


def _oracle_blocks(spec, P, x, masks):
    """The oracle's forward (oracle.model_forward's own ops) with every block's intermediates kept:
    k -> (c, e, a, block input, pooled?)."""
    import torch.nn.functional as F
    S = spec.num_of_patches
    h = F.conv2d(x, P["conv1.weight"], P["conv1.bias"], stride=spec.stem_s, padding=spec.stem_p)
    keep = {}
    for k in range(spec.num_blocks):
        skip = h
        a = F.leaky_relu(F.conv2d(h, P[f"residual_blocks.{k}.conv1.weight"], P[f"residual_blocks.{k}.conv1.bias"], padding=1), 0.2)
        c = F.leaky_relu(F.conv2d(a, P[f"residual_blocks.{k}.conv2.weight"], P[f"residual_blocks.{k}.conv2.bias"], padding=1), 0.2)
        e = c * masks[f"residual_blocks.{k}"][:, :, None, None] + skip
        h = e
        pooled = h.shape[2] > spec.pool_mult * S
        keep[k] = (c, e, a, skip, pooled)
        if pooled:
            h = F.max_pool2d(h, 2)
    return keep


def _route_bytes(c, e):
    """Routing bytes (include/fdet.h: bits 0-3 c > 0 per window element in scan order, bits 4-5 argmax, first
    maximum wins) from fp32 CPU tensors, and the gap between each window's two largest entries."""
    N, C, H, W = e.shape
    ew = e.unfold(2, 2, 2).unfold(3, 2, 2).reshape(N, C, H // 2, W // 2, 4)
    cw = c.unfold(2, 2, 2).unfold(3, 2, 2).reshape(N, C, H // 2, W // 2, 4)
    arg = ew.argmax(dim=-1)                                   # first maximum (torch CPU argmax returns the first)
    bits = sum(((cw[..., k] > 0).long() << k) for k in range(4))
    top2 = ew.topk(2, dim=-1).values
    return (bits | (arg << 4)).to(torch.uint8), top2[..., 0] - top2[..., 1]


def test_gradient_entries_differ_only_through_pool_routing(fd):
    """at F=64 in the benchmarked bf16x3 arithmetic single gradient entries sit up to 5e-2 of the
    tensor scale away from the fp32 oracle while forward and loss hold 1e-4.  The claim was that only the 2x2 max-pool
    routing (a discontinuous function of the activations) does this.  Proof on the product kernels:
      (1) the routing bytes written by the fused forward differ from the oracle's ONLY in windows whose two largest
          entries are within rounding of each other, and the saved conv1 activations differ in SIGN (the LeakyReLU kink,
          the other discontinuity of the backward pass) only where they are within rounding of zero;
      (2) with the oracle's decisions (routing bytes; a and c of every block, which differ from the kernel's by rounding
          only) put in place of the kernel's, EVERY parameter-gradient entry is within 1e-4 of its tensor's scale of the
          oracle's autograd -- per entry, the north star's tolerance (measured: <= 1e-5; with the kernel's own decisions
          up to 4.5e-3, caused by ONE window and a handful of activations at a kink)."""
    from fdet_amd import hotpath as hp
    F_, size, S, B = 64, 480, 10, 3
    spec = O.poolresnet_spec(F_, (3, size, size), S)
    P = O.init_params(spec, seed=3)
    model = _load(_build(fd, "poolresnet", F_, size, S, 10), P).train()
    eng = model.engine
    assert eng.x3 and eng.pool_fusion
    x = torch.rand(B, 3, size, size, generator=torch.Generator().manual_seed(11))
    y = torch.stack([O.encode_targets(b, (size, size), S) for b in O.synthetic_boxes(B, size, seed=1)])
    masks = O.make_dropout_masks(spec, B, seed=101)
    # oracle: gradients by autograd, pooled-block intermediates by the same ops
    leaves = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    loss_ref = O.batch_loss(O.model_forward(spec, leaves, x, masks), y)
    G_ref = dict(zip(leaves, torch.autograd.grad(loss_ref, list(leaves.values()))))
    with torch.no_grad():
        inter = _oracle_blocks(spec, P, x, masks)
    assert [k for k in inter if inter[k][4]] == [0, 1]
    names, params = model.named_stack_params()
    Pd = {n: p.data for n, p in zip(names, params)}
    md = {k: v.cuda() for k, v in masks.items()}

    def run(replace_routes):
        y_hat, saved = eng.forward(x.cuda(), Pd, md, save=True)
        _, lsum, dy = hp.yolo_loss_fwd_bwd(y_hat, y.cuda(), want_grad=True)
        stats = {}
        for k, (c_ref, e_ref, a_ref, xin_ref, pooled) in inter.items():
            xin, a, third = saved["blocks"][k]
            if hasattr(a, "to_f32"):                         # pre-split (PS) path: bf16 hi|lo activation, channel-innermost routing bytes
                a = a.to_f32()
            if hasattr(third, "to_f32"):                     # chain blocks keep c's hi plane only: its signs are what backward reads
                third = third.to_f32()
            if third.dtype == torch.uint8 and third.dim() == 5:
                n_, g_, hp_, wp_, _ = third.shape
                third = third.permute(0, 1, 4, 2, 3).reshape(n_, g_ * 8, hp_, wp_)
            a_got = a.cpu()
            # LeakyReLU kinks: elements whose SIGN the kernel and the oracle disagree on (conv1's a; conv2's c below)
            flip_a = (a_got > 0) != (a_ref > 0)
            worst_kink = float(a_ref[flip_a].abs().max()) / float(a_ref.abs().max()) if flip_a.any() else 0.0
            n_flip = int(flip_a.sum())
            assert float((a_got - a_ref).abs().max()) <= 1e-4 * float(a_ref.abs().max())
            if pooled:
                assert third.dtype == torch.uint8                # the fused forward kept routing bytes, not c
                ref_bytes, gap = _route_bytes(c_ref, e_ref)
                got = third.cpu()
                diff_arg = ((got >> 4) != (ref_bytes >> 4))
                n_flip += int(((got & 15) != (ref_bytes & 15)).sum())
                stats[k] = (int(diff_arg.sum()), float(gap[diff_arg].max()) / float(e_ref.abs().max()) if diff_arg.any() else 0.0,
                            diff_arg.numel(), n_flip, worst_kink)
                if replace_routes:
                    saved["blocks"][k] = (xin, a_ref.cuda(), ref_bytes.cuda())
            else:
                c_got = third.cpu()
                flip_c = (c_got > 0) != (c_ref > 0)
                if flip_c.any():
                    worst_kink = max(worst_kink, float(c_ref[flip_c].abs().max()) / float(c_ref.abs().max()))
                stats[k] = (0, 0.0, a_ref.numel(), n_flip + int(flip_c.sum()), worst_kink)
                if replace_routes:
                    saved["blocks"][k] = (xin, a_ref.cuda(), c_ref.cuda())
        G = {n: torch.empty_like(p) for n, p in Pd.items()}
        eng.backward(saved, dy, Pd, G)
        return float(lsum), {n: v.cpu() for n, v in G.items()}, stats

    lsum, G_own, stats = run(False)
    assert abs(lsum - float(loss_ref)) <= 1e-4 * float(loss_ref)
    for k, (n_arg, worst_gap, total, n_flip, worst_kink) in stats.items():
        # (1): every window routed differently is a near-tie (gap below 1e-4 of the map's scale), every LeakyReLU input
        # whose sign differs is within 1e-4 of zero -- and both are rare
        assert worst_gap <= 1e-4 and worst_kink <= 1e-4, (k, worst_gap, worst_kink)
        assert n_arg <= 2e-4 * total and n_flip <= 1e-3 * total, (k, n_arg, n_flip, total)
    _, G_inj, _ = run(True)
    errs = []
    for n in names:
        ref = G_ref[n].double()
        scale = max(float(ref.abs().max()), 1e-3)
        errs.append((n, float((G_own[n].double() - ref).abs().max()) / scale, float((G_inj[n].double() - ref).abs().max()) / scale))
    print("per-entry gradient error / tensor scale (own routing, oracle routing):")
    for n, eo, ei in errs:
        print(f"   {n:40s} {eo:.2e} {ei:.2e}")
    print(f"windows routed differently: { {k: v[0] for k, v in stats.items() if v[0]} }, lrelu inputs with another sign: "
          f"{ {k: v[3] for k, v in stats.items() if v[3]} }")
    for n, eo, ei in errs:
        assert ei <= 1e-4, (n, ei)                           # (2): per entry, the north star's tolerance (measured <= 1e-5)


def test_gradients_per_entry_in_exact_fp32_arithmetic(fd, monkeypatch):
    """The same statement in the exact-fp32 MFMA arithmetic (FDET_PRECISION=f32; the separate tail kernels recompute the
    pooling argmax from the saved fp32 tensors): even there single entries move by 1e-3 of the scale, because the kernel's
    fmaf chains and oneDNN's differ by summation order (~1e-6) and a tied window / an activation at the LeakyReLU kink
    flips at ANY non-zero difference.  With the oracle's saved tensors (block input, a, c of every block) in place
    of the kernel's, every entry is within 1e-4 -- per entry, the north star's tolerance."""
    monkeypatch.setenv("FDET_PRECISION", "f32")
    from fdet_amd import hotpath as hp
    F_, size, S, B = 64, 480, 10, 2
    spec = O.poolresnet_spec(F_, (3, size, size), S)
    P = O.init_params(spec, seed=3)
    model = _load(_build(fd, "poolresnet", F_, size, S, 10), P).train()
    eng = model.engine
    assert not eng.x3
    x = torch.rand(B, 3, size, size, generator=torch.Generator().manual_seed(11))
    y = torch.stack([O.encode_targets(b, (size, size), S) for b in O.synthetic_boxes(B, size, seed=1)])
    masks = O.make_dropout_masks(spec, B, seed=101)
    leaves = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    loss_ref = O.batch_loss(O.model_forward(spec, leaves, x, masks), y)
    G_ref = dict(zip(leaves, torch.autograd.grad(loss_ref, list(leaves.values()))))
    with torch.no_grad():
        inter = _oracle_blocks(spec, P, x, masks)
    names, params = model.named_stack_params()
    Pd = {n: p.data for n, p in zip(names, params)}

    def run(inject):
        y_hat, saved = eng.forward(x.cuda(), Pd, {k: v.cuda() for k, v in masks.items()}, save=True)
        _, lsum, dy = hp.yolo_loss_fwd_bwd(y_hat, y.cuda(), want_grad=True)
        if inject:
            for k, (c_ref, e_ref, a_ref, xin_ref, _pooled) in inter.items():
                xin, a, c = saved["blocks"][k]
                for got, ref in ((xin, xin_ref), (a, a_ref), (c, c_ref)):
                    assert float((got.cpu() - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max()))
                saved["blocks"][k] = (xin_ref.cuda(), a_ref.cuda(), c_ref.cuda())
        G = {n: torch.empty_like(p) for n, p in Pd.items()}
        eng.backward(saved, dy, Pd, G)
        return float(lsum), {n: v.cpu() for n, v in G.items()}

    lsum, G_own = run(False)
    assert abs(lsum - float(loss_ref)) <= 1e-5 * float(loss_ref)
    _, G_inj = run(True)
    worst_own = worst_inj = 0.0
    for n in names:
        ref = G_ref[n].double()
        scale = max(float(ref.abs().max()), 1e-3)
        worst_own = max(worst_own, float((G_own[n].double() - ref).abs().max()) / scale)
        e_inj = float((G_inj[n].double() - ref).abs().max()) / scale
        worst_inj = max(worst_inj, e_inj)
        assert e_inj <= 1e-4, (n, e_inj)
    print(f"f32 arithmetic, worst per-entry gradient error / tensor scale: own decisions {worst_own:.2e}, oracle's {worst_inj:.2e}")


def test_optimizer_state_dict_resume_equals_uninterrupted_run(fd):
    """a run resumed from (model.state_dict(), optimizer.state_dict()) continues with its Adam moments
    and bias-correction step: two steps in a row == one step, checkpoint round trip into fresh objects, one more step."""
    from fdet_amd.models import ModelMeta
    F_, size, S, B = 16, 480, 10, 2
    spec = O.poolresnet_spec(F_, (3, size, size), S)
    P = O.init_params(spec, seed=7)
    g = torch.Generator().manual_seed(5)
    xs = [torch.rand(B, 3, size, size, generator=g).cuda() for _ in range(2)]
    ys = [torch.stack([O.encode_targets(b, (size, size), S) for b in O.synthetic_boxes(B, size, seed=s)]).cuda() for s in (1, 2)]
    masks = [O.make_dropout_masks(spec, B, seed=s) for s in (11, 12)]

    def fresh():
        model = _load(_build(fd, "poolresnet", F_, size, S, 10), P).train()
        mm = ModelMeta(model=model, lr=1e-4)
        mm.configure_optimizers()
        return model, mm

    model_a, mm_a = fresh()
    for i in range(2):
        model_a.set_dropout_masks(masks[i]); mm_a.fused_train_step(xs[i], ys[i])
    model_b, mm_b = fresh()
    model_b.set_dropout_masks(masks[0]); mm_b.fused_train_step(xs[0], ys[0])
    ck_model = {k: v.detach().cpu().clone() for k, v in model_b.state_dict().items()}
    ck_opt = mm_b.opt.state_dict()
    assert float(ck_opt["state"][0]["step"]) == 1.0 and len(ck_opt["state"]) == len(list(model_b.parameters()))
    model_c, mm_c = fresh()
    model_c.load_state_dict(ck_model)
    mm_c.opt.load_state_dict(ck_opt)
    assert mm_c.opt.step_count == 1
    model_c.set_dropout_masks(masks[1]); mm_c.fused_train_step(xs[1], ys[1])
    for (n, pa), (_, pc) in zip(model_a.named_parameters(), model_c.named_parameters()):
        assert torch.equal(pa.detach().cpu(), pc.detach().cpu()), n          # same kernels, same inputs: bit for bit
    # a resume WITHOUT the optimizer state restarts the bias correction: visibly different parameters
    model_d, mm_d = fresh()
    model_d.load_state_dict(ck_model)
    model_d.set_dropout_masks(masks[1]); mm_d.fused_train_step(xs[1], ys[1])
    assert any(not torch.equal(pa.detach().cpu(), pd.detach().cpu())
               for (_, pa), (_, pd) in zip(model_a.named_parameters(), model_d.named_parameters()))


def test_torchscript_export_matches_eager(fd, golden, tmp_path):
    """SURVEY.md 8f rank 4 (train_model.py:61, convert_checkpoint_to_scripted_model.py:51-54): the scripted module
    (custom fdet:: operators) gives the eager model's maps and the eager demo-path boxes, before and after a
    torch.jit.save / load round trip, with the shipped small PoolResnet weights."""
    from fdet_amd.models.PoolResnet import PoolResnet
    g = golden("g6_trained_small")
    P = {k[len("param/"):]: v for k, v in g.items() if k.startswith("param/")}
    model = PoolResnet(filters=32, input_shape=(3, 480, 480), num_of_patches=10, probability_threshold=0.7,
                       iou_threshold=0.01)
    model = _load(model, P).eval()
    path = tmp_path / "small_scripted.pt"
    scripted = model.to_torchscript(str(path))
    loaded = torch.jit.load(str(path))
    for n in range(g["images"].shape[0]):
        u8 = g["images"][n].cuda()
        pair = torch.stack([u8, u8])
        with torch.no_grad():
            y_e = model(pair.float() / 255.0)
            det_e = model(pair, predict=torch.tensor(1))
            for sm in (scripted, loaded):
                assert torch.equal(sm(pair.float() / 255.0), y_e)
                det_s = sm(pair, torch.tensor(1))
                assert det_s.shape == det_e.shape and torch.equal(det_s.cpu(), det_e.cpu())
        nd = int(g["ndets"][n])                       # ... which are the reference's own detections (g6)
        assert det_e.shape[0] == nd
        if nd:
            assert torch.equal(det_e[:, 1:].cpu(), g["dets"][n, :nd, 1:])
    # torchvision.ops.nms as scripted reference code calls it resolves to the HIP kernel (torchvision is absent here)
    boxes = torch.tensor([[0., 0., 10., 10.], [1., 1., 11., 11.], [20., 20., 30., 30.]]).cuda()
    scores = torch.tensor([0.9, 0.8, 0.7]).cuda()
    assert torch.ops.fdet.nms(boxes, scores, 0.5).tolist() == [0, 2]
    if hasattr(torch.ops, "torchvision") and hasattr(torch.ops.torchvision, "nms"):
        assert torch.ops.torchvision.nms(boxes, scores, 0.5).tolist() == [0, 2]


def test_fused_train_step_is_bit_reproducible(fd):
    """Every reduction of the step (weight-gradient slabs, loss sum, head partials) runs in a fixed order and nothing
    uses float atomics: two models started from the same parameters and fed the same batches and dropout masks hold
    bit-identical parameters, gradients and losses after three steps (F=64: the benchmarked bf16x3 kernels)."""
    from fdet_amd.models import ModelMeta
    F, size, S, nb, B = 64, 480, 10, 10, 4
    spec = O.poolresnet_spec(F, (3, size, size), S, nb)
    P = O.init_params(spec, seed=5)
    runs = []
    for _ in range(2):
        model = _load(_build(fd, "poolresnet", F, size, S, nb), P)
        mm = ModelMeta(model=model, lr=1e-4)
        mm.configure_optimizers()
        model.train()
        g = torch.Generator().manual_seed(17)
        losses = []
        for step in range(1, 4):
            x = torch.rand(B, 3, size, size, generator=g).cuda()
            y = torch.stack([O.encode_targets(b, (size, size), S) for b in O.synthetic_boxes(B, size, seed=step)]).cuda()
            model.set_dropout_masks(O.make_dropout_masks(spec, B, seed=200 + step))
            lsum, y_hat, _ = mm.fused_train_step(x, y)
            losses.append(float(lsum))
        runs.append((losses, mm.opt.space.flat.clone(), mm.opt.space.grad.clone(), y_hat.clone()))
    assert runs[0][0] == runs[1][0]
    for a, b in zip(runs[0][1:], runs[1][1:]):
        assert torch.equal(a, b)


def test_full_batch_pre_split_path_equals_fp32_io_path(fd, monkeypatch):
    """At the benchmarked size (256 images, PoolResnet-medium; the oracle cannot run there in seconds) the engine's two
    data paths are checked against each other: pre-split activations + PS kernels (default) and fp32 NCHW between all
    kernels (FDET_PS=0, the round-2 path that the small-batch tests pin to the oracle).  Same parameters, batch and dropout
    masks: the loss agrees to 1e-6 relative, the maps to 1e-5, every gradient tensor to 5e-4 in relative L2 (both paths feed
    the MFMAs the same hi/lo bf16 pairs and differ in summation order only -- and in the pooling routes / LeakyReLU signs
    that flip at such differences), and the result does not depend on which images
    share a batch (image 0's maps in the full batch == in a batch of 3)."""
    from fdet_amd.models import ModelMeta
    F, size, S, nb, B = 64, 480, 10, 10, 256
    spec = O.poolresnet_spec(F, (3, size, size), S, nb)
    P = O.init_params(spec, seed=7)
    g = torch.Generator().manual_seed(23)
    x = torch.rand(B, 3, size, size, generator=g).cuda()
    y = torch.stack([O.encode_targets(b, (size, size), S) for b in O.synthetic_boxes(B, size, seed=3)]).cuda()
    masks = O.make_dropout_masks(spec, B, seed=77)
    out = {}
    for ps_on in ("1", "0"):
        monkeypatch.setenv("FDET_PS", ps_on)
        model = _load(_build(fd, "poolresnet", F, size, S, nb), P)
        assert model.engine.ps == (ps_on == "1")
        mm = ModelMeta(model=model, lr=1e-4)
        mm.configure_optimizers()
        model.train()
        y3 = y3i = None
        if ps_on == "1":                                     # (before the step: Adam changes the parameters)
            model.set_dropout_masks({k: v[:3] for k, v in masks.items()})
            with torch.no_grad():
                y3i = model(x[:3].contiguous()).clone()      # inference-style forward: the fp32 VALU head (fdet_head_fwd)
            twin = _load(_build(fd, "poolresnet", F, size, S, nb), P).train()
            twin.set_dropout_masks({k: v[:3] for k, v in masks.items()})
            mm3 = ModelMeta(model=twin, lr=1e-4)
            mm3.configure_optimizers()
            _, y3, _ = mm3.fused_train_step(x[:3].contiguous(), y[:3].contiguous())   # the same path as the full batch
            y3 = y3.clone()
            del twin, mm3
        model.set_dropout_masks(masks)
        lsum, y_hat, _ = mm.fused_train_step(x, y)
        names, params = model.named_stack_params()
        out[ps_on] = (float(lsum), y_hat.clone(), mm.opt.space.grad.clone(), names, [p.shape for p in params])
        if y3 is not None:
            assert torch.equal(y3, y_hat[:3])                # bit for bit: the result does not depend on the batch
            # the training step's head is the fused bf16x3 kernel (fdet_head_loss_fused), the plain forward's the fp32 one
            assert float((y3i - y_hat[:3]).abs().max()) <= 1e-5
    (la, ya, ga, names, shapes), (lb, yb, gb, _, _) = out["1"], out["0"]
    assert abs(la - lb) <= 1e-6 * abs(lb), (la, lb)
    assert float((ya - yb).abs().max()) <= 1e-5
    off = 0
    for n, shp in zip(names, shapes):
        k = int(torch.tensor(shp).prod())
        a, b = ga[off:off + k], gb[off:off + k]
        off += k
        # (not 1e-4 per entry: over 256 images a handful of pooling windows are near-ties and a handful of LeakyReLU inputs
        #  sit within rounding of zero; the two paths round differently there, and one flipped route moves single entries
        #  by ~1e-3 of the scale -- test_gradient_entries_differ_only_through_pool_routing pins that mechanism)
        assert float((a - b).abs().max()) <= 2e-3 * max(1e-6, float(b.abs().max())), n
        assert float((a - b).norm()) <= 5e-4 * max(1e-6, float(b.norm())), n
