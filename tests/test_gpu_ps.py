"""GPU parity of the pre-split (PS) activation path (csrc/fdet_ps.h, fdet_conv3x3_ps.hip): format round trip and the
PS conv kernels against a plain PyTorch fp32 CPU reference of the same op (tolerance 1e-4 of the tensor's scale, as
everywhere; the PS format itself keeps 16 significant bits: |x - (hi+lo)| <= 2^-16 |x|)."""
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


# (N, C, H, W): the PoolResnet maps, odd / short maps, a single image, more images than one band run covers
PS_SHAPES = [(3, 64, 60, 60), (5, 64, 30, 30), (1, 64, 60, 60), (2, 64, 12, 56), (9, 64, 16, 30), (2, 64, 7, 17)]


@pytest.mark.parametrize("shape", PS_SHAPES + [(2, 16, 15, 15), (1, 8, 3, 5)])
def test_ps_round_trip_and_zero_halos(env, shape):
    hp, ps = env
    N, C, H, W = shape
    g = torch.Generator().manual_seed(H * 100 + W)
    x = torch.randn(N, C, H, W, generator=g) * torch.exp(torch.randn(N, C, H, W, generator=g) * 3)
    x[0, 0, 0, 0] = 0.0
    t = ps.PsTensor.from_f32(x.cuda())
    y = t.to_f32().cpu()
    assert torch.all((y - x).abs() <= x.abs() * 2.0 ** -16)
    # everything that is not a real element is still zero: sum of |bits| over the buffer == sum over the real elements
    raw = t.buf.view(torch.int16).cpu()
    nz = int((raw != 0).sum())
    t2 = ps.PsTensor(N, C, H, W, "cuda")
    ps.PsTensor.from_f32(torch.ones(N, C, H, W, device="cuda"), out=t2)
    real = int((t2.buf.view(torch.int16).cpu() != 0).sum())          # ones: hi != 0, lo == 0 -> one word per element
    assert real == N * C * H * W
    assert nz <= 2 * N * C * H * W


@pytest.mark.parametrize("shape", PS_SHAPES)
def test_conv3x3_ps_fwd_and_dgrad(env, shape):
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
    xr = xp.to_f32().cpu()                                   # the input as the PS format holds it
    yp = ps.PsTensor(N, C, H, W, "cuda")
    ps.conv3x3_ps_fwd(xp, wf, b.cuda(), yp, slope=0.2)
    z = F.leaky_relu(F.conv2d(xr, w, b, padding=1), 0.2)
    close(yp.to_f32(), z)
    # the halos of the OUTPUT are untouched zeros (a consumer relies on them)
    real = ps.PsTensor.from_f32(torch.full((N, C, H, W), 1.0 + 2.0 ** -9, device="cuda"))    # hi and lo both non-zero
    outside = real.buf.view(torch.int16) == 0
    assert int((yp.buf.view(torch.int16)[outside] != 0).sum()) == 0
    # data gradient with the LeakyReLU derivative of a saved activation
    dz = torch.randn(N, C, H, W, generator=g)
    act = torch.randn(N, C, H, W, generator=g)
    dzp = ps.PsTensor.from_f32(dz.cuda()); ap = ps.PsTensor.from_f32(act.cuda())
    dxp = ps.PsTensor(N, C, H, W, "cuda")
    ps.conv3x3_ps_dgrad_act(dzp, wb, ap, dxp, slope=0.2)
    ref = F.conv_transpose2d(dzp.to_f32().cpu(), w, padding=1) * torch.where(act > 0, 1.0, 0.2)
    close(dxp.to_f32(), ref)


@pytest.mark.parametrize("shape", PS_SHAPES)
@pytest.mark.parametrize("L", [1, 2])
def test_conv3x3_wgrad_ps(env, shape, L):
    hp, ps = env
    N, C, H, W = shape
    g = torch.Generator().manual_seed(N * 77 + H + W + L)
    xs = [torch.randn(N, C, H, W, generator=g) for _ in range(L)]
    dzs = [torch.randn(N, C, H, W, generator=g) for _ in range(L)]
    xp = [ps.PsTensor.from_f32(x.cuda()) for x in xs]
    zp = [ps.PsTensor.from_f32(z.cuda()) for z in dzs]
    dW = [torch.full((C, C, 3, 3), float("nan"), device="cuda") for _ in range(L)]
    db = [torch.full((C,), float("nan"), device="cuda") for _ in range(L)]
    nb = ps.conv3x3_wgrad_ps_ws_bytes(L, N, C, H, W)
    assert nb > 0
    ws = torch.empty(nb // 4, device="cuda")
    ps.conv3x3_wgrad_ps_batched(xp, zp, dW, db, ws)
    for l in range(L):
        xr, zr = xp[l].to_f32().cpu().double(), zp[l].to_f32().cpu().double()
        ref_w = torch.nn.grad.conv2d_weight(xr, (C, C, 3, 3), zr, padding=1)
        close(dW[l], ref_w)
        close(db[l], zr.sum(dim=(0, 2, 3)))
    # bit-reproducible
    dW2 = [torch.empty_like(t) for t in dW]; db2 = [torch.empty_like(t) for t in db]
    ps.conv3x3_wgrad_ps_batched(xp, zp, dW2, db2, ws)
    assert all(torch.equal(a, b) for a, b in zip(dW + db, dW2 + db2))


def _route_to_nchw(route8):
    N, G, Hp, Wp, _ = route8.shape
    return route8.permute(0, 1, 4, 2, 3).reshape(N, G * 8, Hp, Wp)


POOL_SHAPES = [(3, 64, 60, 60), (5, 64, 30, 30), (2, 64, 12, 56), (9, 64, 16, 30)]


@pytest.mark.parametrize("shape", POOL_SHAPES)
@pytest.mark.parametrize("train", [True, False])
def test_conv3x3_ps_pooled_block(env, shape, train):
    """Forward tail fused into conv2 and both backward pieces against torch ops on the CPU; routing bytes checked
    wherever the reference leaves no doubt (sign of c away from zero, a unique window maximum)."""
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
    pool_ps = ps.PsTensor(N, C, H // 2, W // 2, "cuda")
    pool_f = torch.full((N, C, H // 2, W // 2), float("nan"), device="cuda")
    route = ps.route8_like(N, C, H, W, "cuda") if train else None
    ps.conv3x3_ps_fwd_pool(xp, wf, b.cuda(), sp, scale.cuda() if train else None, pool_ps, pool_f, route)
    xr, sr = xp.to_f32().cpu(), sp.to_f32().cpu()
    c = F.leaky_relu(F.conv2d(xr, w, b, padding=1), 0.2)
    u = c * (scale[:, :, None, None] if train else 1.0) + sr
    ref, idx = F.max_pool2d(u, 2, return_indices=True)
    close(pool_f, ref)
    close(pool_ps.to_f32(), ref)
    if not train:
        return
    rt = _route_to_nchw(route).cpu().int()
    # argmax: windows whose best two values differ clearly
    win = F.unfold(u.reshape(N * C, 1, H, W), 2, stride=2).reshape(N, C, 4, H // 2, W // 2)   # scan order (r0c0, r0c1, r1c0, r1c1)
    top2 = win.topk(2, dim=2).values
    clear = (top2[:, :, 0] - top2[:, :, 1]) > 1e-3
    assert torch.equal(((rt >> 4) & 3)[clear], win.argmax(dim=2).int()[clear])
    cw = F.unfold(c.reshape(N * C, 1, H, W), 2, stride=2).reshape(N, C, 4, H // 2, W // 2)
    for k in range(4):
        sure = cw[:, :, k].abs() > 1e-3
        assert torch.equal(((rt >> k) & 1)[sure], (cw[:, :, k] > 0).int()[sure])
    # backward of the tail from the routing bytes: dz2 = unpool(dout) * scale * lrelu'(c)
    dout = torch.randn(N, C, H // 2, W // 2, generator=g)
    dz2 = ps.PsTensor(N, C, H, W, "cuda")
    ps.pool_route_bwd_ps(dout.cuda(), route, scale.cuda(), dz2)
    arg = (rt >> 4) & 3
    ref_dz = torch.zeros(N, C, 4, H // 2, W // 2)
    gs = dout * scale[:, :, None, None]
    for k in range(4):
        ref_dz[:, :, k] = torch.where(arg == k, gs * torch.where(((rt >> k) & 1) == 1, 1.0, 0.2), torch.zeros(()))
    ref_dz = F.fold(ref_dz.reshape(N * C, 4, -1), (H, W), 2, stride=2).reshape(N, C, H, W)
    got = dz2.to_f32().cpu()
    assert torch.all((got - ref_dz).abs() <= ref_dz.abs() * 2.0 ** -15)
    # conv1's data gradient + the un-pooled skip gradient
    dz1 = torch.randn(N, C, H, W, generator=g)
    dzp = ps.PsTensor.from_f32(dz1.cuda())
    dx = torch.full((N, C, H, W), float("nan"), device="cuda")
    ps.conv3x3_ps_dgrad_unpool(dzp, wb, dout.cuda(), route, dx)
    un = torch.zeros(N, C, 4, H // 2, W // 2)
    for k in range(4):
        un[:, :, k] = torch.where(arg == k, dout, torch.zeros(()))
    un = F.fold(un.reshape(N * C, 4, -1), (H, W), 2, stride=2).reshape(N, C, H, W)
    close(dx, F.conv_transpose2d(dzp.to_f32().cpu(), w, padding=1) + un)


@pytest.mark.parametrize("N", [1, 3])
def test_stem_fwd_ps(env, N):
    hp, ps = env
    g = torch.Generator().manual_seed(N)
    x = torch.rand(N, 3, 480, 480, generator=g)
    w = torch.randn(64, 3, 10, 10, generator=g) * 0.05
    b = torch.randn(64, generator=g)
    y = ps.PsTensor(N, 64, 60, 60, "cuda")
    ps.stem_fwd_ps(x.cuda(), w.cuda(), b.cuda(), y, 10, 8, 2)
    ref = F.conv2d(x, w, b, stride=8, padding=2)
    close(y.to_f32(), ref)
    real = ps.PsTensor.from_f32(torch.full((N, 64, 60, 60), 1.0 + 2.0 ** -9, device="cuda"))
    assert int((y.buf.view(torch.int16)[real.buf.view(torch.int16) == 0] != 0).sum()) == 0
