"""GPU parity of the PS kernels on COLUMN STRIPS (csrc/fdet_ps.h; round 4): maps wider than 63 columns -- config 3's 320 / 160 /
80-column levels (models/Resnet.py:30-40) -- kept as strips of <= 62 columns whose edge slots hold the neighbour strip's
column.  Every check is against plain PyTorch fp32 CPU ops on the operands as the PS format holds them (tolerance 1e-4 of
the tensor's scale, as in test_gpu_ps.py)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import fdet_amd
    from fdet_amd import hotpath, ps
    return hotpath, ps


def close(a, b, tol=1e-4):
    a = a.cpu().double(); b = b.cpu().double()
    scale = max(1.0, float(b.abs().max()))
    err = float((a - b).abs().max())
    assert err <= tol * scale, f"max err {err} vs scale {scale}"


# (N, C, H, W): two strips of equal width, three strips with a narrower last one, six strips (the 320 level's plan), one image
STRIP_SHAPES = [(2, 64, 12, 80), (3, 64, 10, 160), (1, 64, 8, 320), (2, 64, 6, 126), (1, 64, 16, 64)]


def test_strip_plan(env):
    hp, ps = env
    assert ps.strips_of(60) == (1, 60) and ps.strips_of(63) == (1, 63)
    assert ps.strips_of(64) == (2, 32)
    assert ps.strips_of(80) == (2, 40)
    assert ps.strips_of(160) == (3, 54)
    assert ps.strips_of(320) == (6, 54)
    assert ps.strips_of(65)[0] == 0                         # odd and wide: no layout


@pytest.mark.parametrize("shape", STRIP_SHAPES)
def test_strip_round_trip_halos_and_exchange(env, shape):
    hp, ps = env
    N, C, H, W = shape
    g = torch.Generator().manual_seed(H * 100 + W)
    x = torch.randn(N, C, H, W, generator=g)
    t = ps.PsTensor.from_f32(x.cuda())
    assert t.strips == ps.strips_of(W)[0] > 1
    y = t.to_f32().cpu()
    assert torch.all((y - x).abs() <= x.abs() * 2.0 ** -16)
    # the halos from_f32 wrote == what halo_exchange writes; zero_only clears exactly them
    full = t.buf.clone()
    ps.halo_exchange(t, zero_only=True)
    cleared = t.buf.clone()
    assert not torch.equal(full, cleared)
    assert torch.equal(t.to_f32().cpu(), y)
    ps.halo_exchange(t)
    assert torch.equal(t.buf, full)
    S, Ws = ps.strips_of(W)
    n_halo = 2 * (S - 1) * N * C * H                          # elements (each a hi and a lo bf16)
    diff = int((full.view(torch.int16) != cleared.view(torch.int16)).sum())
    assert 0 < diff <= 2 * n_halo


@pytest.mark.parametrize("shape", STRIP_SHAPES)
def test_strip_conv_fwd_and_dgrad(env, shape):
    hp, ps = env
    N, C, H, W = shape
    g = torch.Generator().manual_seed(N * 1000 + H + W)
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(C, C, 3, 3, generator=g) * 0.1
    b = torch.randn(C, generator=g)
    nf, nb = hp.packed_sizes(C, C)
    wf = torch.empty(nf, device="cuda"); wb = torch.empty(nb, device="cuda")
    hp.pack_conv3x3_weights(w.cuda(), wf, wb, x3=True)
    xp = ps.PsTensor.from_f32(x.cuda())
    xr = xp.to_f32().cpu()
    yp = ps.PsTensor(N, C, H, W, "cuda")
    ps.conv3x3_ps_fwd(xp, wf, b.cuda(), yp, slope=0.2)
    z = F.leaky_relu(F.conv2d(xr, w, b, padding=1), 0.2)
    close(yp.to_f32(), z)
    # a second conv on the output: needs the exchange (without it the strip edges see zeros)
    y2 = ps.PsTensor(N, C, H, W, "cuda")
    ps.halo_exchange(yp)
    ps.conv3x3_ps_fwd(yp, wf, b.cuda(), y2, slope=0.2)
    z2 = F.leaky_relu(F.conv2d(yp.to_f32().cpu(), w, b, padding=1), 0.2)
    close(y2.to_f32(), z2)
    # data gradient
    dz = torch.randn(N, C, H, W, generator=g)
    act = torch.randn(N, C, H, W, generator=g)
    dzp = ps.PsTensor.from_f32(dz.cuda()); ap = ps.PsTensor.from_f32(act.cuda())
    dxp = ps.PsTensor(N, C, H, W, "cuda")
    ps.conv3x3_ps_dgrad_act(dzp, wb, ap, dxp, slope=0.2)
    ref = F.conv_transpose2d(dzp.to_f32().cpu(), w, padding=1) * torch.where(act > 0, 1.0, 0.2)
    close(dxp.to_f32(), ref)


@pytest.mark.parametrize("shape", STRIP_SHAPES)
def test_strip_wgrad(env, shape):
    hp, ps = env
    N, C, H, W = shape
    g = torch.Generator().manual_seed(N * 77 + H + W)
    x = torch.randn(N, C, H, W, generator=g)
    dz = torch.randn(N, C, H, W, generator=g)
    xp = ps.PsTensor.from_f32(x.cuda())                      # x: real halos
    zp = ps.PsTensor.from_f32(dz.cuda())
    ps.halo_exchange(zp, zero_only=True)                     # dz: zero halos (a halo slot is not a position of its strip)
    dW = [torch.full((C, C, 3, 3), float("nan"), device="cuda")]
    db = [torch.full((C,), float("nan"), device="cuda")]
    nb = ps.conv3x3_wgrad_ps_ws_bytes(1, N, C, H, W)
    assert nb > 0
    ws = torch.empty(nb // 4, device="cuda")
    ps.conv3x3_wgrad_ps_batched([xp], [zp], dW, db, ws)
    xr, zr = xp.to_f32().cpu().double(), zp.to_f32().cpu().double()
    close(dW[0], torch.nn.grad.conv2d_weight(xr, (C, C, 3, 3), zr, padding=1))
    close(db[0], zr.sum(dim=(0, 2, 3)))


@pytest.mark.parametrize("shape", [(2, 64, 12, 80), (3, 64, 10, 160), (1, 64, 8, 320), (2, 64, 6, 126)])
@pytest.mark.parametrize("train", [True, False])
def test_strip_pooled_block(env, shape, train):
    """Pooled residual-block tail on strips: forward (fp32 NCHW pooled output + per-strip routing bytes) and both backward
    pieces against torch ops on the CPU."""
    hp, ps = env
    N, C, H, W = shape
    g = torch.Generator().manual_seed(N * 31 + H + W + int(train))
    x = torch.randn(N, C, H, W, generator=g)
    skip = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(C, C, 3, 3, generator=g) * 0.1
    b = torch.randn(C, generator=g)
    scale = (torch.rand(N, C, generator=g) > 0.25).float() / 0.75 if train else None
    nf, nb = hp.packed_sizes(C, C)
    wf = torch.empty(nf, device="cuda"); wb = torch.empty(nb, device="cuda")
    hp.pack_conv3x3_weights(w.cuda(), wf, wb, x3=True)
    xp = ps.PsTensor.from_f32(x.cuda()); sp = ps.PsTensor.from_f32(skip.cuda())
    pool_f = torch.full((N, C, H // 2, W // 2), float("nan"), device="cuda")
    route = ps.route8_like(N, C, H, W, "cuda") if train else None
    ps.conv3x3_ps_fwd_pool(xp, wf, b.cuda(), sp, scale.cuda() if train else None, None, pool_f, route)
    xr, sr = xp.to_f32().cpu(), sp.to_f32().cpu()
    c = F.leaky_relu(F.conv2d(xr, w, b, padding=1), 0.2)
    u = c * (scale[:, :, None, None] if train else 1.0) + sr
    ref, idx = F.max_pool2d(u, 2, return_indices=True)
    close(pool_f, ref)
    if not train:
        return
    # backward through the routing bytes: dz2 = unpool(dout) * scale * lrelu'(c); dx = conv^T(dz1) + unpool(dout)
    dout = torch.randn(N, C, H // 2, W // 2, generator=g)
    dz2 = ps.PsTensor.from_f32(torch.randn(N, C, H, W, generator=g).cuda())      # a recycled buffer: data AND halo slots dirty
    ps.pool_route_bwd_ps(dout.cuda(), route, scale.cuda(), dz2, slope=0.2)
    after = dz2.buf.view(torch.int32).clone()                  # the kernel cleared the halo slots itself (the weight gradient reads
    ps.halo_exchange(dz2, zero_only=True)                      # dz2 next): the zero pass changes nothing
    assert torch.equal(dz2.buf.view(torch.int32), after)
    un = F.max_unpool2d(dout, idx, 2, output_size=(H, W))
    ref_dz2 = un * scale[:, :, None, None] * torch.where(c > 0, 1.0, 0.2)
    got = dz2.to_f32().cpu()
    sure = (c.abs() > 1e-3)                                  # the sign of c is certain
    srt = u.reshape(N, C, H // 2, 2, W // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(N, C, H // 2, W // 2, 4).sort(dim=-1).values
    uniq = ((srt[..., 3] - srt[..., 2]) > 1e-3).repeat_interleave(2, 2).repeat_interleave(2, 3)   # a unique window maximum
    m = sure & uniq
    assert m.float().mean() > 0.9
    assert float(((got - ref_dz2).abs() * m).max()) <= 1e-4 * max(1.0, float(ref_dz2.abs().max()))
    dz1 = torch.randn(N, C, H, W, generator=g)
    dzp = ps.PsTensor.from_f32(dz1.cuda())
    dx = torch.full((N, C, H, W), float("nan"), device="cuda")
    ps.conv3x3_ps_dgrad_unpool(dzp, wb, dout.cuda(), route, dx, slope=0.2)
    ref_dx = F.conv_transpose2d(dzp.to_f32().cpu(), w, padding=1) + un
    err = ((dx.cpu() - ref_dx).abs() * uniq).max()
    assert float(err) <= 1e-4 * max(1.0, float(ref_dx.abs().max()))
    assert bool(torch.isfinite(dx).all())


def _resnet_step(mode: str, strips: bool):
    """One fused training step of Resnet-64 at 3x256x256 (levels 128 -> 64 -> 32 -> 16 -> 8: strip levels 128 and 64), fixed
    parameters / masks / batch; returns (loss, y, flat gradient)."""
    import oracle as O
    from fdet_amd import hotpath as hp
    from fdet_amd.models import ModelMeta
    from fdet_amd.models.Resnet import Resnet
    size, S, B = 256, 8, 3
    spec = O.resnet_spec(64, (3, size, size), S, 6)
    P = O.init_params(spec, seed=4)
    model = Resnet(filters=64, input_shape=(3, size, size), num_of_patches=S, num_of_residual_blocks=6)
    model.load_state_dict({k: v.clone() for k, v in P.items()})
    model = model.cuda().train()
    assert model.engine.ps and model.engine.ps_strips
    model.engine.ps_strips = strips                          # (what FDET_PS_STRIPS=0 selects: the round-2 kernels for wide maps)
    if mode == "bf16":
        model.engine.set_precision("bf16")
    mm = ModelMeta(model=model, lr=1e-4); mm.configure_optimizers()
    x = torch.rand(B, 3, size, size, generator=torch.Generator().manual_seed(1)).cuda()
    y = hp.encode_targets(O.synthetic_boxes(B, size, seed=2), (size, size), S)
    model.set_dropout_masks({k: v.cuda() for k, v in O.make_dropout_masks(spec, B, seed=3).items()})
    lsum, y_hat, _ = mm.fused_train_step(x, y)
    return float(lsum), y_hat.clone(), mm.opt.space.grad.clone()


def test_strip_levels_equal_the_round2_kernels_end_to_end():
    """The same training step with the wide levels on column strips (PS kernels) and on the round-2 fp32-I/O kernels: both are
    fp32-grade bf16x3 paths, so loss, outputs and the flat gradient agree to the usual 1e-4 / L2 5e-3."""
    la, ya, ga = _resnet_step("bf16x3", True)
    lb, yb, gb = _resnet_step("bf16x3", False)
    assert abs(la - lb) <= 1e-4 * max(1.0, abs(lb)), (la, lb)
    assert float((ya - yb).abs().max()) <= 1e-4
    rel = float((ga - gb).norm() / gb.norm())
    assert rel <= 5e-3, rel


def test_strip_levels_precision16():
    """precision16 (one bf16 MFMA pass, hi planes only -- halo exchange included) on the strip levels against the fp32-grade
    step: bf16-level agreement (tolerances of test_gpu_p16.py)."""
    la, ya, ga = _resnet_step("bf16x3", True)
    lb, yb, gb = _resnet_step("bf16", True)
    assert abs(la - lb) <= 2e-2 * abs(la), (la, lb)
    assert float((ya - yb).abs().max()) <= 2e-2
    cos = float((ga * gb).sum() / (ga.norm() * gb.norm()))
    assert cos >= 0.995, cos
