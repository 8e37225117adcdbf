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


@pytest.mark.parametrize("shape", PS_SHAPES + [(3, 64, 15, 15), (4, 64, 10, 10), (2, 64, 5, 31)])
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


@pytest.mark.parametrize("N", [1, 3])
@pytest.mark.parametrize("p16", [False, True])
def test_stem_fwd_ps_on_uint8_frames(env, N, p16):
    """fdet_stem_fwd_ps_u8: the stem reading the uint8 frames (x / 255 through its 256-entry hi | lo table) writes exactly the
    bytes of fdet_u8_to_f32_norm + fdet_stem_fwd_ps -- every value 0..255 occurs, an all-zero and an all-255 frame row included."""
    hp, ps = env
    g = torch.Generator().manual_seed(10 + N)
    fr = torch.randint(0, 256, (N, 3, 480, 480), dtype=torch.uint8, generator=g)
    fr[0, :, 0] = 0
    fr[0, :, 1] = 255
    fr[0, 0, 2, :256] = torch.arange(256, dtype=torch.uint8)
    w = torch.randn(64, 3, 10, 10, generator=g) * 0.05
    b = torch.randn(64, generator=g)
    fd = fr.cuda()
    y_ref = ps.PsTensor(N, 64, 60, 60, "cuda")
    ps.stem_fwd_ps(hp.u8_to_f32_norm(fd), w.cuda(), b.cuda(), y_ref, 10, 8, 2, p16=p16)
    y = ps.PsTensor(N, 64, 60, 60, "cuda")
    ps.stem_fwd_ps(fd, w.cuda(), b.cuda(), y, 10, 8, 2, p16=p16)
    assert torch.equal(y.buf.view(torch.int32), y_ref.buf.view(torch.int32))
    close(y.to_f32(), F.conv2d(fr.float() / 255.0, w, b, stride=8, padding=2), 1e-4 if not p16 else 2e-2)


@pytest.mark.parametrize("cfg", [(3, 15, 15, 3, True), (2, 10, 10, 2, False), (5, 15, 15, 8, True), (1, 7, 9, 1, True)])
def test_block_chain_ps_flavour(env, cfg):
    """The LDS-resident block chain keeping its per-block tensors in PS (fdet_block_chain_{fwd,bwd}_ps) against torch CPU
    fp32 of models/PoolResnet.py:33-43 (pool == 1) and its autograd, and against the fp32-NCHW flavour of the same kernel:
    the chain output and input gradient are bit-identical, a kept PS tensor is the fp32 one rounded to hi + lo."""
    hp, ps = env
    N, H, W, nb, use_scale = cfg
    C = 64
    g = torch.Generator().manual_seed(N * 100 + H + nb)
    x = torch.randn(N, C, H, W, generator=g)
    Ws = [(torch.randn(C, C, 3, 3, generator=g) * 0.05, torch.randn(C, generator=g) * 0.1,
           torch.randn(C, C, 3, 3, generator=g) * 0.05, torch.randn(C, generator=g) * 0.1) for _ in range(nb)]
    scales = [((torch.rand(N, C, generator=g) > 0.25).float() / 0.75) for _ in range(nb)] if use_scale else None
    x_ps = ps.PsTensor.from_f32(x.cuda())
    xq = x_ps.to_f32()                                      # the input as PS holds it
    xr = xq.cpu().clone().requires_grad_(True)
    h = xr
    a_ref, c_ref, o_ref, z1, z2 = [], [], [], [], []
    for k in range(nb):
        w1, b1, w2, b2 = Ws[k]
        p1 = F.conv2d(h, w1, b1, padding=1); p1.retain_grad(); z1.append(p1)
        a = F.leaky_relu(p1, 0.2)
        p2 = F.conv2d(a, w2, b2, padding=1); p2.retain_grad(); z2.append(p2)
        c = F.leaky_relu(p2, 0.2)
        h = (c * scales[k][:, :, None, None] if use_scale else c) + h
        a_ref.append(a.detach()); c_ref.append(c.detach()); o_ref.append(h.detach())
    dout = torch.randn(N, C, H, W, generator=g)
    h.backward(dout)
    nf, nbk = hp.packed_sizes(C, C)
    wf1, wb1, wf2, wb2 = [], [], [], []
    for (w1, b1, w2, b2) in Ws:
        for w, lf, lb in ((w1, wf1, wb1), (w2, wf2, wb2)):
            f_ = torch.empty(nf, device="cuda"); b_ = torch.empty(nbk, device="cuda")
            hp.pack_conv3x3_weights(w.cuda(), f_, b_, x3=True)
            lf.append(f_); lb.append(b_)
    b1s = [w[1].cuda() for w in Ws]; b2s = [w[3].cuda() for w in Ws]
    sc_d = [s_.cuda() for s_ in scales] if use_scale else None
    mkps = lambda n_: [ps.PsTensor(N, C, H, W, "cuda") for _ in range(n_)]
    mk = lambda: [torch.full((N, C, H, W), float("nan"), device="cuda") for _ in range(nb)]
    # fp32 flavour on the same input
    a_d, c_d, o_d = mk(), mk(), mk()
    hp.block_chain_fwd(xq, wf1, b1s, wf2, b2s, sc_d, a_d, c_d, o_d)
    for x_in in (x_ps, xq):                                  # PS input, fp32 input
        a_p, c_p, o_p = mkps(nb), mkps(nb), mkps(nb - 1)
        last = torch.full((N, C, H, W), float("nan"), device="cuda")
        ps.block_chain_fwd_ps(x_in, wf1, b1s, wf2, b2s, sc_d, a_p, c_p, o_p, last)
        assert torch.equal(last, o_d[-1])
        close(last, o_ref[-1])
        for k in range(nb):
            close(a_p[k].to_f32(), a_ref[k])
            assert torch.equal(a_p[k].to_f32(), ps.PsTensor.from_f32(a_d[k]).to_f32())
            chi = c_p[k].to_f32().cpu()                      # hi plane only
            firm = c_ref[k].abs() > 1e-3 * max(1.0, float(c_ref[k].abs().max()))
            assert torch.equal((chi > 0)[firm], (c_ref[k] > 0)[firm])
            assert torch.equal(chi > 0, c_d[k].cpu() > 0)
            if k + 1 < nb:
                close(o_p[k].to_f32(), o_ref[k])
                assert torch.equal(o_p[k].to_f32(), ps.PsTensor.from_f32(o_d[k]).to_f32())
        # nothing outside the real elements was written
        real = ps.PsTensor.from_f32(torch.full((N, C, H, W), 1.0 + 2.0 ** -9, device="cuda"))
        outside = real.buf.view(torch.int16) == 0
        for t in a_p + c_p + o_p:
            assert int((t.buf.view(torch.int16)[outside] != 0).sum()) == 0
    # inference flavour: nothing kept
    last2 = torch.full((N, C, H, W), float("nan"), device="cuda")
    ps.block_chain_fwd_ps(x_ps, wf1, b1s, wf2, b2s, sc_d, None, None, None, last2)
    assert torch.equal(last2, o_d[-1])
    # backward, fp32 dout and dx.  The kept activations are the REFERENCE's here (as in test_block_chain_fwd_bwd): a
    # device-computed a / c within 1e-4 of zero may carry the other sign, and one flipped LeakyReLU' is a visible gradient
    # difference that says nothing about the backward kernel.
    a_t = [ps.PsTensor.from_f32(t.cuda()) for t in a_ref]; c_t = [ps.PsTensor.from_f32(t.cuda()) for t in c_ref]
    dz1_p, dz2_p = mkps(nb), mkps(nb)
    dx_p = torch.full((N, C, H, W), float("nan"), device="cuda")
    ps.block_chain_bwd_ps(dout.cuda(), wb1, wb2, sc_d, a_t, c_t, dz1_p, dz2_p, dx_p)
    dz1_d, dz2_d = mk(), mk()
    dx_d = torch.full((N, C, H, W), float("nan"), device="cuda")
    hp.block_chain_bwd(dout.cuda(), wb1, wb2, sc_d, [t.cuda() for t in a_ref], [t.cuda() for t in c_ref], dz1_d, dz2_d, dx_d)
    # ... and with the PS tensors the forward kept: the two flavours agree bit for bit on those as well
    dz1_q, dz2_q = mkps(nb), mkps(nb)
    dx_q = torch.full((N, C, H, W), float("nan"), device="cuda")
    ps.block_chain_bwd_ps(dout.cuda(), wb1, wb2, sc_d, a_p, c_p, dz1_q, dz2_q, dx_q)
    dx_e = torch.full((N, C, H, W), float("nan"), device="cuda")
    hp.block_chain_bwd(dout.cuda(), wb1, wb2, sc_d, a_d, c_d, mk(), mk(), dx_e)
    assert torch.equal(dx_q, dx_e)
    assert torch.equal(dx_p, dx_d)
    close(dx_p, xr.grad)
    for k in range(nb):
        close(dz1_p[k].to_f32(), z1[k].grad)
        close(dz2_p[k].to_f32(), z2[k].grad)
        assert torch.equal(dz1_p[k].to_f32(), ps.PsTensor.from_f32(dz1_d[k]).to_f32())
        assert torch.equal(dz2_p[k].to_f32(), ps.PsTensor.from_f32(dz2_d[k]).to_f32())


@pytest.mark.parametrize("shape", [(3, 60, 60), (5, 30, 30), (2, 12, 56)])
def test_conv3x3_ps_two_chunk_layers(env, shape):
    """Layers with 32 channels on the contraction side (two 16-channel chunks per tile instead of four: the woven epilogue
    then shares BOTH chunks with the next tile): forward 32 -> 64 channels and the data gradient of a 64 -> 32 layer."""
    hp, ps = env
    N, H, W = shape
    g = torch.Generator().manual_seed(N * 31 + H + W)
    # forward, Cin = 32, Cout = 64
    x = torch.randn(N, 32, H, W, generator=g)
    w = torch.randn(64, 32, 3, 3, generator=g) * 0.1
    b = torch.randn(64, generator=g)
    nf, nb = hp.packed_sizes(64, 32)
    wf = torch.empty(nf, device="cuda"); wb = torch.empty(nb, device="cuda")
    hp.pack_conv3x3_weights(w.cuda(), wf, wb, x3=True)
    xp = ps.PsTensor.from_f32(x.cuda())
    yp = ps.PsTensor(N, 64, H, W, "cuda")
    ps.conv3x3_ps_fwd(xp, wf, b.cuda(), yp, slope=0.2)
    close(yp.to_f32(), F.leaky_relu(F.conv2d(xp.to_f32().cpu(), w, b, padding=1), 0.2))
    # data gradient of a layer with Cin = 64, Cout = 32: dx (64 channels) = conv^T(dz (32 channels)) * lrelu'(act)
    w2 = torch.randn(32, 64, 3, 3, generator=g) * 0.1
    nf2, nb2 = hp.packed_sizes(32, 64)
    wf2 = torch.empty(nf2, device="cuda"); wb2 = torch.empty(nb2, device="cuda")
    hp.pack_conv3x3_weights(w2.cuda(), wf2, wb2, x3=True)
    dz = torch.randn(N, 32, H, W, generator=g)
    act = torch.randn(N, 64, H, W, generator=g)
    dzp = ps.PsTensor.from_f32(dz.cuda()); ap = ps.PsTensor.from_f32(act.cuda())
    dxp = ps.PsTensor(N, 64, H, W, "cuda")
    ps.conv3x3_ps_dgrad_act(dzp, wb2, ap, dxp, slope=0.2)
    ref = F.conv_transpose2d(dzp.to_f32().cpu(), w2, padding=1) * torch.where(act > 0, 1.0, 0.2)
    close(dxp.to_f32(), ref)
